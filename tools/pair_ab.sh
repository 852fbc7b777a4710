#!/bin/bash
# in-situ A/B of the fused conv3 -> conv1 pairs: whole bench (network + tail), interleaved, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for f in 0 1; do echo -n "fuse_pair=$f r$rep: "; AIRPOSE_FUSE_PAIR=$f python bench.py --steps 20 --warmup 5 --cpu-sample 0 --parity-steps 0 --repeat-blocks 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f pairs/s  %.3f ms/step  conv %.3f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['stage_ms_per_step']['conv_stack'], d['roofline']['frac']))"; done; done
