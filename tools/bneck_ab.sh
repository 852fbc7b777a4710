#!/bin/bash
# stand-alone A/B of two library builds of the fused layer1 bottleneck on one box, interleaved: tools/bneck_ab.sh <suffix> [reps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in $(seq 1 ${2:-2}); do for suf in "$1" ""; do for ds in 0 1; do echo -n "lib${suf:-(product)} r$rep: "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip$suf.so python tools/bneck_bench.py --images 512 --cuts 2 --ds $ds 2>&1 | grep cut; done; done; done
