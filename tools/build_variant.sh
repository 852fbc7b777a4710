#!/bin/bash
# A/B build of ONE 16-bit source with extra defines: tools/build_variant.sh block_img e16 -DBI_EXP=16 -> airpose_amd/libairpose_hip_e16.so
# (the other objects come from the last `make`; select the library with AIRPOSE_HIP_LIB)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/airpose_amd/csrc
SRC=$1; SUF=$2; shift 2
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-local-typedef"
cd $C
OBJS=$(ls *.o | grep -v trace | grep -v "^$SRC\.")
/opt/rocm/bin/hipcc $FL "$@" -c $SRC.hip -o /tmp/${SRC}_$SUF.o &
/opt/rocm/bin/hipcc $FL "$@" -DAP_F16 -c $SRC.hip -o /tmp/${SRC}_$SUF.f16.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/airpose_amd/libairpose_hip_$SUF.so $OBJS /tmp/${SRC}_$SUF.o /tmp/${SRC}_$SUF.f16.o -Wl,-rpath,/opt/rocm/lib
echo built libairpose_hip_$SUF.so
