#!/bin/bash
# rocprofv3 kernel-trace durations of the stem + pool kernel in a one-pass bench run (512 images per launch), for each value of
# AIRPOSE_STEM given: tools/stem_trace.sh 1 2        (1: persistent form, 2: a workgroup per strip)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/st_$v
  AIRPOSE_STEM=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$v -- python $R/bench.py --steps 5 --warmup 2 \
    --parity-sweep 0 --airpose-plus 0 --b64 0 --cpu-sample 0 --parity-steps 0 --parity-pairs 0 --repeat-blocks 0 --stage-steps 0 \
    --other-form 0 --dual-stream ${DUAL:-0} ${EXTRA} > /tmp/st_$v.log 2>&1
  f=$(find /tmp/st_$v -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then echo "AIRPOSE_STEM=$v:"; grep -i "stem" "$f" | sed -e 's/(float const[^"]*"/"/' | cut -c1-160; else echo "AIRPOSE_STEM=$v: no trace"; tail -3 /tmp/st_$v.log; fi
done
