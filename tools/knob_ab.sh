#!/bin/bash
# whole-bench A/B of one environment knob of bench.py, interleaved on one box: tools/knob_ab.sh AIRPOSE_FUSE_POOL "0 1" [reps] [extra bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
K=$1; VALS=$2; REPS=${3:-3}; shift 3
for rep in $(seq 1 $REPS); do for v in $VALS; do echo -n "$K=$v r$rep: "; env $K=$v python bench.py --steps 20 --warmup 5 --cpu-sample 0 --parity-steps 0 --b64 0 --repeat-blocks 2 --repeat-steps 50 --airpose-plus 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('%.0f pairs/s  %.3f ms/step  conv %.3f ms  frac %.4f  stem %.3f avgpool %.3f' % (d['repeat_blocks']['median'], d['ms_per_step'], s['conv_stack'], d['roofline']['frac'], s['stem_maxpool'], s['avgpool']))"; done; done
