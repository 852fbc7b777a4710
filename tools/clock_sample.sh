#!/bin/bash
# GPU clock / power once a second while the bench loop runs -> gpurun_out/clock_sample.txt (rocm-smi; the microarchitecture
# guide's 2.4 GHz peak clock is what bench.py prices against: this shows what the chip actually holds under the trunk)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
OUT=gpurun_out/clock_sample.txt; : > $OUT
python bench.py --steps ${1:-4000} --warmup 20 --cpu-sample 0 --parity-steps 0 --b64 0 --repeat-blocks 0 --stage-steps 0 > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
t=0
while kill -0 $BP 2>/dev/null; do
  s=$(rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power \(W\)" | sed -e 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz)/sclk \1 MHz/' -e 's/.*Power (W): \([0-9.]*\)/power \1 W/' | tr '\n' ' ')
  echo "t=$t $s" >> $OUT; t=$((t+1)); sleep 1
done
python -c "import json; d=json.loads(open('gpurun_out/clock_bench.json').read().strip().splitlines()[-1]); print('bench: %.0f pairs/s, %.3f ms/step, frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))" >> $OUT
sort -t' ' -k3,3n $OUT | awk '{print $3}' | sort -n | uniq -c | sort -k2,2n | tail -12
tail -45 $OUT
