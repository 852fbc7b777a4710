#!/bin/bash
# A/B of two library builds on the SMPL-X tail (tools/lbs_bench.py) and on the whole bench: airpose_amd/libairpose_hip_base.so
# (the previous build, copied there by hand) against the product library.   bash tools/probes/lbs_lib_ab.sh [full]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
 for L in libairpose_hip_base.so libairpose_hip.so; do
  echo -n "$L r$rep: "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/$L python tools/lbs_bench.py --bodies 512,4096 --iters 30 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print([(r['bodies'], round(r['ms']*1e3,1), {k:round(v*1e3,1) for k,v in r.items() if k.endswith('_ms')}) for r in d['rows']])" 2>&1 | tail -1
 done
done
if [ "$1" = "full" ]; then
 for rep in 1 2; do
  for L in libairpose_hip_base.so libairpose_hip.so; do
   echo -n "$L bench r$rep: "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/$L python bench.py --cpu-sample 0 --parity-steps 0 --b64 0 --repeat-blocks 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), d['repeat_blocks']['median'], {k:(round(v,4) if isinstance(v,float) else v) for k,v in d['stage_ms_per_step'].items() if k!='source'}, round(d['smplx_tail_roofline']['frac'],4))"
  done
 done
fi
