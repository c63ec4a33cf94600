#!/bin/bash
# round 6, second half: the final tree - kernel stats (100 steps), PMC traffic, the bench line, 512 x 512, the transitions alone
bash scripts/profile_round.sh
python bench.py > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err; tail -c 600 gpurun_out/r06h_bench.json
python scripts/bench_512.py --out gpurun_out/r06h_bench_512.json > /dev/null 2>&1; grep -E "frames_per_s|ms_per_batch" gpurun_out/r06h_bench_512.json
python scripts/kbench.py --kernels tr --iters 50 2>&1 | grep "'k'" > gpurun_out/r06h_kbench_tr.txt; python scripts/kbench.py --kernels tr --iters 50 --ws 2>&1 | grep "'k'" >> gpurun_out/r06h_kbench_tr.txt; cat gpurun_out/r06h_kbench_tr.txt
