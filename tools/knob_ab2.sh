#!/bin/bash
# interleaved A/B of one environment knob of bench.py on one box: tools/knob_ab2.sh AIRPOSE_IMG3 0 1 [reps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
K=$1; A=$2; B=$3; R=${4:-3}
for rep in $(seq $R); do
for v in $A $B; do
  export $K=$v
  echo -n "$K=$v: "
  python bench.py --steps 20 --warmup 5 --parity-sweep 0 --airpose-plus 0 --b64 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 2 --repeat-steps 60 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.0f pairs/s (blocks median %.0f) | stream-ordered %.0f | conv span %.3f ms | launches %d' % (d['value'], d['repeat_blocks']['median'], d['stream_ordered']['pairs_per_s'], d['stream_ordered']['conv_stack_ms'], d['roofline']['launches_per_step']))"
done; done
