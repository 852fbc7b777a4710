#!/bin/bash
# LDS bank-conflict share of every kernel of a bench step (one PMC pass, counters only): tools/pmc_lds_all.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_lds_all; rm -rf $O; mkdir -p $O
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $O/pass -- python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --parity-sweep 0 --airpose-plus 0 --b64 0 --parity-steps 0 --parity-pairs 0 --repeat-blocks 0 --stage-steps 0 --other-form 0 --dual-stream 0 ${EXTRA} > $O/pass.log 2>&1 || true
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sorted(glob.glob("$O/pass/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(anonymous namespace\)::", "", r.get("Kernel_Name", ""))
        k = re.sub(r"\(.*", "", k)[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-72s %6s %12s %12s %8s %10s" % ("kernel", "n", "LDS active", "conflict", "share", "LDS/MFMA"))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_LDS_IDX_ACTIVE", [0]))):
    a = sum(d.get("SQ_LDS_IDX_ACTIVE", [0])); c = sum(d.get("SQ_LDS_BANK_CONFLICT", [0])); n = len(d.get("SQ_LDS_IDX_ACTIVE", []))
    m = sum(d.get("SQ_INSTS_MFMA", [0])); l = sum(d.get("SQ_INSTS_LDS", [0]))
    if a < 1e5: continue
    print("%-72s %6d %12.4g %12.4g %7.1f%% %10.3f" % (k, n, a / max(n, 1), c / max(n, 1), 100 * c / a, l / m if m else 0))
PY
