#!/bin/bash
# timing-only builds of the fused contraction + skinning kernel (smplx.hip, -DLF_ABLATE=<bits>; libairpose_hip_labl<bits>.so):
#   1 no vertex stores | 2 no skinning | 4 no fragment refills | 8 no MFMAs
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python tools/lbs_bench.py --bodies 512 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1])['rows'][0]; print('total %.1f us | prep %.1f  fused/blend %.1f  skin %.1f  joints %.1f' % (r['ms']*1e3, r['prep_ms']*1e3, r['blend_gemm_ms']*1e3, r['skin_ms']*1e3, r['joints_ms']*1e3))"; }
echo -n "product:        "; run
for a in ${LBS_ABL:-1 2 4 8 7}; do echo -n "LF_ABLATE=$a:   "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/libairpose_hip_labl$a.so run; done
