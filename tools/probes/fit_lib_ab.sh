#!/bin/bash
# A/B of two library builds on the AirPose+ fitting loop (BASELINE config 5): libairpose_hip_base.so (the previous build, copied
# there by hand) against the product library.   bash tools/probes/fit_lib_ab.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
 for L in libairpose_hip_base.so libairpose_hip.so; do
  echo -n "$L r$rep: "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/$L python tools/fit_bench.py --cpu-iters 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(round(d['value'],3),'ms', round(d['ms_per_iteration']*1e3,2),'us/iter', 'loss', d['loss_first'], d['loss_last'])"
 done
done
