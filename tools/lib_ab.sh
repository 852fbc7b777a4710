#!/bin/bash
# A/B of library builds on the same box: tools/lib_ab.sh "<lib suffixes, '' = product>" "<layers>" [cfg]
cd /root/repo
CFG=${3:-11}
for rep in 1 2; do for suf in $1; do L=airpose_amd/libairpose_hip$suf.so; [ "$suf" = "base" ] && L=airpose_amd/libairpose_hip.so
  for o in $2; do echo -n "$suf r$rep "; AIRPOSE_HIP_LIB=$PWD/$L python tools/conv_bench.py --images 512 --only $o --cfgs=$CFG --iters 30 2>&1 | grep "$o"; done; done; done
