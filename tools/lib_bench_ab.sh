#!/bin/bash
# whole-bench A/B of library builds, interleaved on one box: tools/lib_bench_ab.sh "<suffixes, '-' = product>" [reps] [extra bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
SUFS=$1; REPS=${2:-3}; shift 2
for rep in $(seq 1 $REPS); do for suf in $SUFS; do s=$suf; [ "$suf" = "-" ] && s=""; echo -n "lib${s:-(product)} r$rep: "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip$s.so python bench.py --steps 20 --warmup 5 --cpu-sample 0 --parity-steps 0 --b64 0 --repeat-blocks 2 --repeat-steps 50 --airpose-plus 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('%.0f pairs/s  %.3f ms/step  conv %.3f ms  frac %.4f' % (d['repeat_blocks']['median'], d['ms_per_step'], s['conv_stack'], d['roofline']['frac']))"; done; done
