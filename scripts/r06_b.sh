#!/bin/bash
set -x
mkdir -p gpurun_out
python tests/tools/parity_timed.py --out gpurun_out/r06b_parity_timed.json > gpurun_out/r06b_parity.log 2>&1
grep -v "^   " gpurun_out/r06b_parity.log | tail -6
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_golden.py tests/test_gpu_kernels.py tests/test_gpu_calibration.py -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-exact-line | tail -1 > gpurun_out/r06b_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06b_bench.json')); print(d['value'], d['roofline']['families_ms_per_step']); print(d['config']['parity_live'])"
