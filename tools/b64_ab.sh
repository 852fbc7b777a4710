#!/bin/bash
# BASELINE config 1 size (64 pairs, network only): A/B of the size rules of the image-resident kernels: tools/b64_ab.sh "1 1" "2 1" "1 2" "2 2"  (IMG_BLOCK IMG3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for v in "$@"; do
  set -- $v
  export AIRPOSE_IMG_BLOCK=$1 AIRPOSE_IMG3=$2
  echo -n "img_block=$1 img3=$2: "
  python bench.py --batch 64 --no-tail --steps 40 --warmup 5 --parity-sweep 0 --airpose-plus 0 --b64 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 2 --repeat-steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.0f pairs/s (blocks median %.0f) | launches %d' % (d['value'], d['repeat_blocks']['median'], d['roofline']['launches_per_step']))"
  set -- "$@"
done; done
