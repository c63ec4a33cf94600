#!/bin/bash
# scripts/power_round.sh <tag>: power / clock traces (scripts/power_trace.py) of the kernels DESIGN.md calls power-bound, into gpurun_out/
tag=${1:-r04}
mkdir -p gpurun_out
python scripts/power_trace.py --out gpurun_out/${tag}_power_mfma_only.json --seconds 6 --settle 3 -- ./scripts/microbench/mfmabench loop 400
python scripts/power_trace.py --out gpurun_out/${tag}_power_strip56.json --seconds 6 --settle 8 -- python scripts/kbench.py --kernels ds --blocks 0 --iters 400000
python scripts/power_trace.py --out gpurun_out/${tag}_power_block14.json --seconds 6 --settle 8 -- python scripts/kbench.py --kernels b14 --iters 400000
python scripts/power_trace.py --out gpurun_out/${tag}_power_block14_b128.json --seconds 6 --settle 8 -- python scripts/kbench.py --kernels b14 --iters 400000 --batch 128
python scripts/power_trace.py --out gpurun_out/${tag}_power_bench.json --seconds 8 --settle 25 -- python bench.py --steps 20000 --no-cpu-baseline --no-exact-line --single-region
