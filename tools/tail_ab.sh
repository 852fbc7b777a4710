#!/bin/bash
# in-bench A/B of the SMPL-X tail (instrumented stream-ordered steps behind the trunk): tools/tail_ab.sh 9 1   (AIRPOSE_SMPLX_FUSED values)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
for v in "$@"; do
  export AIRPOSE_SMPLX_FUSED=$v
  echo -n "smplx_fused=$v: "
  python bench.py --steps 10 --warmup 3 --stage-steps 30 --parity-sweep 0 --airpose-plus 0 --b64 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --other-form 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
s = d['stage_ms_per_step']
print('prep %.1f  fused %.1f  joints %.1f us | tail frac %.4f | %.0f pairs/s' % (1e3*s['smplx_prep'], 1e3*s.get('smplx_lbs_fused', 0), 1e3*s['smplx_joints'], d['smplx_tail_roofline']['frac'], d['value']))"
done; done
