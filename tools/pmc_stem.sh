R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_stem; rm -rf $O; mkdir -p $O
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $O/pass$i -- timeout 300 python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-tail --parity-sweep 0 --airpose-plus 0 --b64 0 --parity-steps 0 --parity-pairs 0 --repeat-blocks 0 --stage-steps 0 --other-form 0 --dual-stream 0 ${EXTRA} > $O/pass$i.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$O/pass*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        k = r.get("Kernel_Name", "")
        if "stem" not in k: continue
        acc[k[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k)
        for c, v in d.items(): print("   %-28s avg %.4g  (n=%d)" % (c, sum(v)/len(v), len(v)))
PY
