#!/bin/bash
# timing-only builds of the second-cut bottleneck (bottleneck2.hip, -DB2_ABLATE=<bits>; libairpose_hip_babl<bits>.so next to
# the product library): 1 no x loads after the prologue | 2 no stores | 4 no W3 DMA | 8 no MFMAs | 16 no epilogue arithmetic
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== product build"; python tools/bneck_bench.py --images ${IMAGES:-512}
for a in ${B2_ABL:-1 2 3 4 8 16 27}; do echo "B2_ABLATE=$a:"; AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_babl$a.so python tools/bneck_bench.py --images ${IMAGES:-512} --cuts 2; done
