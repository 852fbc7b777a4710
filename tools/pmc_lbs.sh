#!/bin/bash
# SQ counter passes (no tracing) over the fused SMPL-X contraction + skinning kernel at 512 bodies; run on the GPU box from the repo
# root: tools/pmc_lbs.sh [fused: 1 second cut | 3 first cut]
R=$PWD; F=${1:-1}; cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/ps; timeout 200 rocprofv3 --pmc $P --output-format csv -d /tmp/ps -- python $R/tools/lbs_bench.py --bodies 512 --iters 3 --fused $F > /tmp/ps.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for p in glob.glob("/tmp/ps/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "lbs" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("%-26s avg %.4g (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
done
