python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for e in TN_NO_TRANS_WS=1 TN_X=1; do echo $e; env $e python scripts/bench_512.py --out gpurun_out/bench_512_$e.json 2>&1 | grep -E "frames_per_s|transition" -A1 | head -8; done
