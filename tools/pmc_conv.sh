#!/bin/bash
# PMC passes (counters only, no tracing) for one conv shape/config: tools/pmc_conv.sh <only> <cfg> [images]
# writes gpurun_out/pmc_<only>_<cfg>/pass*/ and prints the per-kernel averages.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; ONLY=$1; CFG=$2; IMG=${3:-512}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_${ONLY}_${CFG}; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python $R/tools/conv_bench.py --images $IMG --only $ONLY --cfgs=$CFG --iters 5 > $OUT/pass$i.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r.get("Kernel_Name", "")
        if "conv_" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print("   %-28s avg %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
