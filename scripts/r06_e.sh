export TMPDIR=/tmp; R=$PWD; cd /tmp
FT_BATCH=64 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ft -- python $R/scripts/prof_finetune.py > $R/gpurun_out/prof_ft.log 2>&1
cd $R
f=$(find gpurun_out/prof_ft -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06e_finetune_kernel_stats_b64.csv; head -40 $f | cut -c1-160
find gpurun_out/prof_ft -name "*kernel_trace.csv" -delete
