#!/bin/bash
# Vector-memory-path counters (TA / TCP / TCC) of one conv shape and config: tools/pmc_ta.sh <only> <cfg> [images]
# counters only, one rocprofv3 --pmc pass per group; prints per-kernel averages.
R=${GRAFT_REPO_ROOT:-/root/repo}; ONLY=$1; CFG=$2; IMG=${3:-512}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmcta_${ONLY}_${CFG}; mkdir -p $OUT
i=0
for P in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python $R/tools/conv_bench.py --images $IMG --only $ONLY --cfgs=$CFG --iters 5 > $OUT/pass$i.log 2>&1 || tail -3 $OUT/pass$i.log
done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r.get("Kernel_Name", "")
        if "conv_" not in k: continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print("   %-40s avg %.5g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
