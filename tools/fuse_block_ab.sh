#!/bin/bash
# whole-bench A/B of the layer1 bottleneck cuts (AIRPOSE_FUSE_BLOCK = 1 first cut, 2 second cut), interleaved: tools/fuse_block_ab.sh [reps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq 1 ${1:-3}); do for fb in 1 2; do echo -n "fuse_block=$fb r$rep: "; AIRPOSE_FUSE_BLOCK=$fb python bench.py --steps 20 --warmup 5 --cpu-sample 0 --parity-steps 0 --repeat-blocks 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('%.0f pairs/s  %.3f ms/step  conv %.3f ms  frac %.4f' % (d['value'], d['ms_per_step'], s['conv_stack'], d['roofline']['frac']))"; done; done
