python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "1x1 or transition" 2>&1 | tail -5
for rep in 1 2 3; do
echo old; python scripts/kbench.py --kernels tr --iters 50 --blocks 1,2 2>&1 | grep "'k'"
echo ws; python scripts/kbench.py --kernels tr --iters 50 --blocks 1,2 --ws 2>&1 | grep "'k'"
done
