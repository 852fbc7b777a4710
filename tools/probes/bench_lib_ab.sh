#!/bin/bash
# Interleaved A/B of two library builds on the whole bench: airpose_amd/libairpose_hip_base.so (the previous build, copied there by
# hand) against the product library.   bash tools/probes/bench_lib_ab.sh [reps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq 1 ${1:-3}); do
 for L in libairpose_hip_base.so libairpose_hip.so; do
  echo -n "$L r$rep: "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/$L python bench.py --cpu-sample 0 --parity-steps 0 --b64 0 --repeat-blocks 4 --stage-steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['repeat_blocks']['median']), {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['stage_ms_per_step'].items() if k!='source'})"
 done
done
