for rep in 1 2 3; do
echo new; python scripts/kbench.py --kernels tr --iters 50 2>&1 | grep "'k'"
echo old; python scripts/kbench.py --kernels tr --iters 50 --lib _ab/libtennis_c1old.so 2>&1 | grep "'k'"
done
