R=$PWD; export TMPDIR=/tmp
cd /tmp
for l in hip oldswz hip oldswz; do
  export TENNIS_HIP_LIB=$R/tennis_amd/lib/libtennis_$l.so
  rm -rf $R/gpurun_out/ab_$l
  TN_NO_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$l -- python $R/bench.py --no-cpu-baseline --single-region --steps 5 --warmup 2 > $R/gpurun_out/ab_$l.log 2>&1
  echo "lib=$l"; f=$(find $R/gpurun_out/ab_$l -name "*kernel_stats.csv" | head -1); grep -E "dense_layer_kernel<(56|28|14|7)" $f | sed 's/.*dense_layer_kernel<\([0-9]*\)[^"]*",\([0-9]*\),[0-9]*,\([0-9.]*\),.*/\1 \2 \3/'
done
