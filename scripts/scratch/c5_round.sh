mkdir -p gpurun_out
python -m pytest tests/test_gpu_captioner.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/t.log
python scripts/time_c5.py 2>&1 | grep C5 >> gpurun_out/t.log
CELL=lstm python scripts/time_c5.py 2>&1 | grep C5 >> gpurun_out/t.log
python scripts/scratch/dec_stamps.py 2>&1 | tail -4 >> gpurun_out/t.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o c5 --output-format csv -- python $R/scripts/prof_c5.py > /tmp/prof.log 2>&1
find /tmp/prof_c5 -name "*kernel_stats*" -exec cp {} $R/gpurun_out/c5_kernel_stats.csv \;
