#!/usr/bin/env python
"""Turn the raw outputs of tools/collect_profiles.sh <tag> (gpurun_out/<tag>_*) into the small summaries committed
under profiles/: kernel stats csv, per-layer table, HBM traffic csv + json (the json feeds bench.py's roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import sqlite3
import subprocess
import sys

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
O = os.path.join(R, "gpurun_out")
CONV = ("conv_pipe_kernel", "conv_slab_kernel", "conv_lean_kernel", "conv_pair_kernel", "conv_igemm_kernel", "bneck256_kernel", "bneck64ds_kernel", "bneck2_kernel", "blk_img_kernel",
        "conv_pw_kernel", "conv_img3_kernel")


def is_trunk_conv(name):     # the bf16 conv-stack kernels (the fp32 instances are the blend-shape GEMM)
    return any(k in name for k in CONV) and "<float" not in name and "bsplit_t" not in name


db = glob.glob(os.path.join(O, tag + "_trace", "**", "*_results.db"), recursive=True)
if db:
    c = sqlite3.connect(db[0])
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(os.path.join(O, tag + "_kernel_stats.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0 --dual-stream 0  (%s)\n" % tag)
        f.write("name,calls,total_us,avg_us,pct\n")
        for n, calls, tot, avg, pct in rows:
            f.write('"%s",%d,%.3f,%.3f,%.2f\n' % (n[:110], calls, tot, avg, pct))
    with open(os.path.join(O, tag + "_layer_table.txt"), "w") as f:
        subprocess.run([sys.executable, os.path.join(R, "tools", "layer_profile.py"), db[0], "512"], stdout=f, stderr=subprocess.STDOUT)

db2 = glob.glob(os.path.join(O, tag + "_trace2", "**", "*_results.db"), recursive=True)
if db2:
    c = sqlite3.connect(db2[0])
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(os.path.join(O, tag + "_kernel_stats_two_streams.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0  (%s; default = two concurrent trunk passes: kernel durations overlap)\n" % tag)
        f.write("name,calls,total_us,avg_us,pct\n")
        for n, calls, tot, avg, pct in rows:
            f.write('"%s",%d,%.3f,%.3f,%.2f\n' % (n[:110], calls, tot, avg, pct))

acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
steps = 3                                                   # --steps 2 --warmup 1
for ctr, col in (("FETCH_SIZE", 1), ("WRITE_SIZE", 2)):
    for p in glob.glob(os.path.join(O, "%s_pmc_%s" % (tag, ctr), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] != ctr:
                continue
            a = acc[r["Kernel_Name"]]
            a[col] += float(r["Counter_Value"])
            if col == 1:
                a[0] += 1
if acc:
    fetch = write = launches = 0
    with open(os.path.join(O, tag + "_pmc_hbm_traffic.csv"), "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0  (%d passes of the hot path; %s)\n" % (steps, tag))
        f.write("# units: KB as reported; MI355X_MICROARCH.md: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024, and on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x\n")
        f.write("kernel,launches,fetch_KB_total,write_KB_total,fetch_MB_per_launch_raw,fetch_MB_per_launch_x2,write_MB_per_launch\n")
        for k, (n, fe, wr) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            if n == 0:
                continue
            f.write('"%s",%d,%.0f,%.0f,%.1f,%.1f,%.1f\n' % (k[:90], n, fe, wr, fe / n / 1024, 2 * fe / n / 1024, wr / n / 1024))
            if is_trunk_conv(k):
                fetch += fe; write += wr; launches += n
    if launches:
        lp = launches / steps
        j = {"source": "profiles/%s_pmc_hbm_traffic.csv" % tag, "conv_launches_per_step": lp,
             "fetch_bytes_per_step_raw": fetch * 1024 / steps, "fetch_bytes_per_step_x2": 2 * fetch * 1024 / steps,
             "write_bytes_per_step": write * 1024 / steps,
             "traffic_bytes_per_launch": (2 * fetch + write) * 1024 / steps / lp}
        json.dump(j, open(os.path.join(O, tag + "_pmc_traffic.json"), "w"), indent=1)
        print(json.dumps(j))
# MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over every SIMD of the chip) against GRBM_GUI_ACTIVE (cycles the
# GPU was busy with the dispatch; one value per XCD, summed by rocprofv3 over the 8 XCDs) x 1024 SIMDs
macc = collections.defaultdict(lambda: collections.defaultdict(float))
mcnt = collections.Counter()
for p in glob.glob(os.path.join(O, tag + "_pmc_MFMA", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        macc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            mcnt[r["Kernel_Name"]] += 1
if macc:
    tb = tg = 0.0
    with open(os.path.join(O, tag + "_pmc_mfma_busy.csv"), "w") as f:
        f.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python bench.py --steps 2 --warmup 1 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0 --dual-stream 0  (%s)\n" % tag)
        f.write("# mfma_busy_pct = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); 100 %% = every SIMD's matrix pipe busy for the whole dispatch\n")
        f.write("kernel,launches,mfma_busy_cycles,insts_mfma,gui_active_cycles_per_xcd,mfma_busy_pct\n")
        for k, d in sorted(macc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
            g = d.get("GRBM_GUI_ACTIVE", 0.0) / 8
            if g <= 0:
                continue
            b = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            f.write('"%s",%d,%.0f,%.0f,%.0f,%.2f\n' % (k[:90], mcnt[k], b, d.get("SQ_INSTS_MFMA", 0.0), g, 100 * b / (g * 1024)))
            if is_trunk_conv(k):
                tb += b; tg += g
        if tg:
            f.write('"conv stack (all bf16 conv kernels)",,%.0f,,%.0f,%.2f\n' % (tb, tg, 100 * tb / (tg * 1024)))
            print("conv stack MFMA busy %.2f %%" % (100 * tb / (tg * 1024)))
for f in ("bench", "bench_bf16", "bench_b64", "bench_x2", "bench_fp32"):
    p = os.path.join(O, "%s_%s.json" % (tag, f))
    if os.path.exists(p) and os.path.getsize(p):
        d = json.loads(open(p).read().strip().splitlines()[-1])
        print(f, d["value"], d["unit"], "%.3f ms/step" % d["ms_per_step"], "frac %.3f" % d["roofline"]["frac"])
