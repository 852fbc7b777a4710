# bench A/B: automatic choice (slab kernel on the stride-1 3x3 layers) against the ring kernel everywhere, same box
cd /root/repo; mkdir -p gpurun_out/slab_ab
for rep in 1 2; do for c in -1 -4; do echo "== conv config $c"; AIRPOSE_CONV_CONFIG=$c timeout 300 python bench.py --cpu-sample 0 --parity-steps 0 --repeat-blocks 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stage_ms_per_step'))"; done; done 2>&1 | tee gpurun_out/slab_ab/ab.txt
