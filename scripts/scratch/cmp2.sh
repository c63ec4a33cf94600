P='import sys,json
for l in sys.stdin:
    if l.startswith("{\"metric"):
        d=json.loads(l); print(sys.argv[1], "value", d["value"], "avg_launch_us", d["roofline"]["avg_launch_us"], d["roofline"]["families_ms_per_step"]["dense_layer_fused_28x28"])'
python bench.py --no-cpu-baseline --steps 200 --warmup 10 | python3 -c "$P" "default"
python bench.py --no-cpu-baseline --single-region --steps 200 --warmup 10 | python3 -c "$P" "single-region"
python bench.py --no-cpu-baseline --steps 200 --warmup 10 | python3 -c "$P" "default"
python bench.py --no-cpu-baseline --single-region --steps 200 --warmup 10 | python3 -c "$P" "single-region"
