python bench.py > gpurun_out/r06f_bench.log 2>gpurun_out/r06f_bench.err; tail -1 gpurun_out/r06f_bench.log > gpurun_out/r06f_bench.json
python - <<PY
import json; d=json.load(open('gpurun_out/r06f_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['families_ms_per_step']); print(d['config']['parity_live']); print(d['config'].get('frames_per_sec_forwards_joined'), d['config'].get('exact_weights_frames_per_sec')); print(d['cpu_baseline']['value'])
PY
bash scripts/profile_round.sh
