R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/b64_trace
rocprofv3 --kernel-trace --stats -d $O/b64_trace -- python $R/bench.py --batch 64 --no-tail --steps 5 --warmup 2 --stage-steps 0 --cpu-sample 0 --parity-steps 0 --repeat-blocks 0 --b64 0 --dual-stream 0 --chunk 64 > $O/b64_trace.log 2>&1
DB=$(find $O/b64_trace -name "*results.db" | head -1)
python $R/tools/layer_profile.py $DB 64 > $O/s3_layer_table_64img.txt 2>&1
rm -rf $O/b64_trace
tail -45 $O/s3_layer_table_64img.txt
