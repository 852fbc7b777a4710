#!/bin/bash
# HBM-side bytes of one conv shape/config: tools/pmc_fetch.sh <only> <cfg> [images]  (FETCH_SIZE / WRITE_SIZE in separate passes,
# KB as reported; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x -- see MI355X_MICROARCH.md)
R=${GRAFT_REPO_ROOT:-/root/repo}; ONLY=$1; CFG=$2; IMG=${3:-512}
cd /tmp; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmcf_${ONLY}_${CFG}; mkdir -p $OUT
for P in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=$OUT/$(echo $P | tr ' ' '_'); rm -rf $D
  timeout -k 5 150 rocprofv3 --pmc $P --output-format csv -d $D -- python $R/tools/conv_bench.py --images $IMG --only $ONLY --cfgs=$CFG --iters 3 > $D.log 2>&1 || tail -2 $D.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for p in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        if "conv_" in r.get("Kernel_Name", ""): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in acc.items(): print("   %-24s avg %.5g per dispatch (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
