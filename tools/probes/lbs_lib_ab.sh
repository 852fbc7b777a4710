cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "smplx or pipeline or input_meshes" 2>&1 | tail -3
for rep in 1 2 3; do
 for L in libairpose_hip_base.so libairpose_hip.so; do
  echo -n "$L r$rep: "; AIRPOSE_HIP_LIB=$PWD/airpose_amd/$L python tools/lbs_bench.py --bodies 512,4096 --iters 30 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print([(r.get('bodies'), round(r.get('ms',0)*1e3,1), {k:round(v*1e3,1) for k,v in r.items() if k.endswith('_ms')}) for r in (d['rows'] if isinstance(d,dict) and 'rows' in d else d)])" 2>&1 | tail -1
 done
done
