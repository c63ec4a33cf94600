for e in 31 16; do echo "exp $e"; TN_B7_MSPLIT=1 python scripts/kbench.py --kernels b7 --iters 20 --stamps --lib _ab/libtennis_b7m$e.so 2>&1 | grep -E "totals|'k'|l 0|l 1 |l 8|l15"; done
TN_B7_MSPLIT=1 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "block7" 2>&1 | tail -2
