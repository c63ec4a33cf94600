R=$PWD; export TMPDIR=/tmp; cd /tmp
P='import sys,json
for l in sys.stdin:
    if l.startswith("{\"metric"):
        d=json.loads(l); print(sys.argv[1], "value", d["value"], "avg_launch_us(56x56 family, HIP events)", d["roofline"]["avg_launch_us"])'
TN_NO_SPLIT=1 python $R/bench.py --no-cpu-baseline --single-region --steps 5 --warmup 2 | python3 -c "$P" "plain-5-steps-nosplit"
TN_NO_SPLIT=1 python $R/bench.py --no-cpu-baseline --single-region --steps 200 --warmup 10 | python3 -c "$P" "plain-200-steps-nosplit"
python $R/bench.py --no-cpu-baseline --single-region --steps 200 --warmup 10 | python3 -c "$P" "plain-200-steps-split"
rm -rf $R/gpurun_out/cmpp; TN_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cmpp -- python $R/bench.py --no-cpu-baseline --single-region --steps 5 --warmup 2 2>/dev/null | python3 -c "$P" "rocprof-5-steps-nosplit"
f=$(find $R/gpurun_out/cmpp -name "*kernel_stats.csv" | head -1); grep "dense_layer_kernel<56" $f | awk -F, '{print "rocprof stats 56x56 avg ns:", $(NF-4)}'
rm -rf $R/gpurun_out/cmpp; TN_NO_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cmpp -- python $R/bench.py --no-cpu-baseline --single-region --steps 200 --warmup 10 2>/dev/null | python3 -c "$P" "rocprof-200-steps-nosplit"
f=$(find $R/gpurun_out/cmpp -name "*kernel_stats.csv" | head -1); grep "dense_layer_kernel<56" $f | awk -F, '{print "rocprof stats 56x56 avg ns:", $(NF-4)}'
rm -rf $R/gpurun_out/cmpp
