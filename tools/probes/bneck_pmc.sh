#!/bin/bash
# write-path counters of the fused layer1 kernels (stand-alone bench, 512 images): tools/probes/bneck_pmc.sh
# (every pass under its own timeout: a pass with the TA_* counters aborted inside rocprofv3 and then sat until the box's limit)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bneck_pmc; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  for ds in 0 1; do
    timeout 150 rocprofv3 --pmc $C --output-format csv -d $O/p${i}_ds$ds -- python $R/tools/bneck_bench.py --images 512 --cuts 2 --ds $ds --iters 3 > $O/p${i}_ds$ds.log 2>&1
    python - "$O/p${i}_ds$ds" "$ds" <<'PY'
import csv, glob, sys, collections
d, ds = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bneck2" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]["v"] += float(r["Counter_Value"]); n[(r["Counter_Name"], r["Dispatch_Id"])] += 1
disp = len({k[1] for k in n}) or 1
print("ds=%s  " % ds + "  ".join("%s=%.4g" % (k, v["v"] / disp) for k, v in sorted(acc.items())) if acc else "ds=%s: no rows (%s)" % (ds, d))
PY
  done
done
