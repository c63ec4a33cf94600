#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel stats + one PMC pass per counter, outputs under gpurun_out/.
set -u
R=$PWD; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --single-region"
cd /tmp
TN_NO_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- $B --steps 5 --warmup 2 > $R/gpurun_out/prof_stats.log 2>&1
TN_NO_SPLIT=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_fetch -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_fetch.log 2>&1
TN_NO_SPLIT=1 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_write -- $B --steps 2 --warmup 1 > $R/gpurun_out/prof_write.log 2>&1
cd $R
find gpurun_out/prof_fetch gpurun_out/prof_write -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/prof_stats.log
