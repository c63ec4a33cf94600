mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/full.log 2>&1
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -2 gpurun_out/bench.err >> gpurun_out/full.log
