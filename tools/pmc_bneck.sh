#!/bin/bash
# PMC passes (counters only, no tracing) for the fused bottleneck kernel: tools/pmc_bneck.sh [images]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; IMG=${1:-512}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_bneck; mkdir -p $OUT
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python $R/tools/bneck_bench.py --images $IMG --iters 3 > $OUT/pass$i.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r.get("Kernel_Name", "")
        if "bneck" not in k and "conv_" not in k: continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print("   %-28s avg %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
