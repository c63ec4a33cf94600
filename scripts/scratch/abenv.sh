R=$PWD; export TMPDIR=/tmp; cd /tmp
for w in 768 512 256 768; do
  rm -rf $R/gpurun_out/abw
  TN_STEM_WGS=$w TN_NO_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abw -- python $R/bench.py --no-cpu-baseline --single-region --steps 5 --warmup 2 > /dev/null 2>&1
  f=$(find $R/gpurun_out/abw -name "*kernel_stats.csv" | head -1)
  echo "wgs=$w $(grep stem_pool $f | awk -F, '{print $(NF-4)}')"
done
rm -rf $R/gpurun_out/abw
