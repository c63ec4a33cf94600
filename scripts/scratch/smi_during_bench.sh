#!/bin/bash
# Tuning: sample power / clocks while bench.py runs (is the chip at its power cap?)
python bench.py --steps 8000 --warmup 5 --no-cpu-baseline > gpurun_out/smi_bench.json 2> /dev/null &
BP=$!
for i in $(seq 1 30); do
  echo "t=$i $(rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Package Power|sclk|junction' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -1 gpurun_out/smi_bench.json | cut -c1-160
