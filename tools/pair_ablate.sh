#!/bin/bash
# timing-only builds of the pair kernel (conv_pair.hip, -DPR_ABLATE=<bits>; libairpose_hip_pabl<bits>.so next to the product
# library): what each part of a tile costs.  1 no identity loads | 2 no stores | 4 no weight DMA | 8 no MFMAs | 16 no epilogue math | 32 prologue only | 64 / 128 identity loads / stores as full 128-byte lines (shape experiment)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
echo "== product build"; python tools/pair_bench.py --images 256 --rounds 3
for a in ${PAIR_ABL:-1 2 3 4 8 16 19}; do echo "== PR_ABLATE=$a"; AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_pabl$a.so python tools/pair_bench.py --images 256 --rounds 3 --nocheck | sed 's/| two launches.*//'; done
