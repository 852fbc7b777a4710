#!/bin/bash
# SQ counter passes (no tracing) over the SMPL-X skinning kernel at 4096 bodies; run on the GPU box from the repo root.
R=$PWD; cd /tmp; export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAVES"; do
  rm -rf /tmp/ps; timeout 200 rocprofv3 --pmc $P --output-format csv -d /tmp/ps -- python $R/tools/lbs_bench.py --bodies 4096 --iters 3 > /tmp/ps.log 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for p in glob.glob("/tmp/ps/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "skin" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("%-26s avg %.4g (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
done
