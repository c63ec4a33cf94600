python scripts/dead_debug.py "none,bn0,bn1,bn2,trans" 2>/dev/null | grep "^dead"
python scripts/dead_debug.py "bn1" -1 2>/dev/null | grep "^dead"
python scripts/dead_debug.py "bn1" 1 2>/dev/null | grep "^dead"
python tests/tools/parity_timed.py --out gpurun_out/r06d_parity_timed.json > gpurun_out/r06d_parity.log 2>&1
grep -v "Warning\|W.as_fp16\|plain = \|amdgpu" gpurun_out/r06d_parity.log | cut -c1-150
