#!/bin/bash
# Round artefacts on the GPU box: tools/collect_profiles.sh <tag>   (e.g. r01_e)
# Writes gpurun_out/<tag>_*; tools/summarize_profiles.py <tag> turns them into the files committed under profiles/.
# Counter passes are separate rocprofv3 runs with --pmc only (no tracing), as the MI355X guide prescribes.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r01_x}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python $R/bench.py --steps 20 --warmup 5 --batch 64 --no-tail --cpu-sample 0 --parity-steps 0 --repeat-steps 50 > $O/${TAG}_bench_b64.json 2>> $O/${TAG}_bench.err
python $R/bench.py --steps 20 --warmup 5 --precision bf16 --cpu-sample 0 --parity-steps 0 --repeat-blocks 2 --repeat-steps 50 --b64 0 --airpose-plus 0 > $O/${TAG}_bench_bf16.json 2>> $O/${TAG}_bench.err
python $R/bench.py --steps 10 --warmup 2 --precision bf16x2 --cpu-sample 0 --parity-steps 0 --repeat-blocks 1 --repeat-steps 10 --b64 0 --airpose-plus 0 > $O/${TAG}_bench_x2.json 2>> $O/${TAG}_bench.err
python $R/bench.py --steps 5 --warmup 2 --precision fp32 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0 --airpose-plus 0 > $O/${TAG}_bench_fp32.json 2>> $O/${TAG}_bench.err
rm -rf $O/${TAG}_trace $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
# kernel trace of the single-pass trunk (--dual-stream 0): one kernel at a time, so the per-layer table is attributable;
# the product's default (two concurrent passes) is what the bench lines and the counter passes below run
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -- python $R/bench.py --steps 5 --warmup 2 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0 --airpose-plus 0 --dual-stream 0 > $O/${TAG}_trace.log 2>&1
rm -rf $O/${TAG}_trace2
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace2 -- python $R/bench.py --steps 5 --warmup 2 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0 --airpose-plus 0 > $O/${TAG}_trace2.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_$C -- python $R/bench.py --steps 2 --warmup 1 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0 --airpose-plus 0 > $O/${TAG}_pmc_$C.log 2>&1
done
# MFMA-busy pass (SQ + GRBM counters in one pass: independent blocks).  Single trunk pass (--dual-stream 0): with the two
# concurrent passes of the default a kernel's GRBM_GUI_ACTIVE also counts the cycles it shares the chip with the other pass's
# kernel, and busy / active comes out 4 points low (19 % instead of 23 % for the ring kernel)
rm -rf $O/${TAG}_pmc_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_pmc_MFMA -- python $R/bench.py --steps 2 --warmup 1 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --other-form 0 --airpose-plus 0 --dual-stream 0 > $O/${TAG}_pmc_MFMA.log 2>&1
python $R/tools/summarize_profiles.py $TAG
