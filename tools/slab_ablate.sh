# timing-only builds of the slab kernel (conv_slab.hip, -DSL_ABLATE=<bits>): what each part of its K loop costs
cd /root/repo; mkdir -p gpurun_out/slab_abl
for a in ${SLAB_ABL:-0 1 2 8 3 16}; do L=airpose_amd/libairpose_hip_sabl$a.so; [ $a = 0 ] && L=airpose_amd/libairpose_hip.so; echo "== slab ablate $a"; for o in l2.1.c2 l3.1.c2 l4.1.c2; do AIRPOSE_HIP_LIB=$PWD/$L timeout 120 python tools/conv_bench.py --images 512 --only $o --cfgs=11,14 --iters 20 2>&1 | grep "$o"; done; done 2>&1 | tee gpurun_out/slab_abl/slab_ablate.txt
