#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel stats + one PMC pass per counter, outputs under gpurun_out/.
# The stats pass times 100 steps: with 5 (rounds 1 and 2a-2g) the averages are dominated by the first passes, which run
# 10-15 % slower than the steady state (clocks still ramping: 143.9 us against 128.1 us for the 56x56 layers on one box).
set -u
R=$PWD; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-exact-line --no-parity-live --single-region"
# the PMC passes run the fp16-representable seeded weights (--plain-rounding: the same kernels on the same shapes, no 144 calibration
# forwards through the layer-wise kernels mixing into the per-family averages)
P="$B --plain-rounding"
cd /tmp
TN_NO_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B --steps 100 --warmup 10 > $R/gpurun_out/prof_stats.log 2>&1
TN_NO_SPLIT=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- $P --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
TN_NO_SPLIT=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- $P --steps 2 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
cd $R
find gpurun_out/prof_fetch gpurun_out/prof_write -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/prof_stats.log
