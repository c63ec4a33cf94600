#!/bin/bash
# round 6, first GPU call: parity of the timed path, then the round-4 tree against this tree on the same box (alternating)
set -x
mkdir -p gpurun_out
python tests/tools/parity_timed.py --out gpurun_out/r06a_parity_timed.json > gpurun_out/r06a_parity.log 2>&1
tail -5 gpurun_out/r06a_parity.log
for i in 1 2 3; do
  (cd _ab/r4 && python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-exact-line) | tail -1 > gpurun_out/r06a_ab_r4_$i.json
  python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-exact-line --no-parity-live | tail -1 > gpurun_out/r06a_ab_r5_$i.json
  python - <<PY
import json
for t in ("r4","r5"):
    d=json.load(open("gpurun_out/r06a_ab_%s_$i.json"%t)); print(t, d["value"], d["roofline"]["families_ms_per_step"])
PY
done
