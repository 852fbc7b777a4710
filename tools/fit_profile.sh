#!/bin/bash
# rocprofv3 kernel statistics of the AirPose+ fitting loop (tools/fit_bench.py); run on the GPU box:  tools/fit_profile.sh TAG
TAG=${1:-r01_f}
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fitprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fitprof -o fit -- python $R/tools/fit_bench.py --cpu-iters 1 > $R/gpurun_out/${TAG}_fit_trace.log 2>&1
for f in $(find /tmp/fitprof -name "*kernel_stats.csv"); do cp "$f" $R/gpurun_out/${TAG}_fit_kernel_stats.csv; done
python - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/${TAG}_fit_kernel_stats.csv")):
    print(r["Name"][:64], r["Calls"], r["AverageNs"], r["Percentage"])
PY
