# scripts/scratch/ab_prof.sh "lib1 lib2 ..." 'kernel-regex'  (ON the GPU box): rocprof average duration of the matching kernels per library build
R=$PWD; export TMPDIR=/tmp
cd /tmp
for l in $1; do
  export TENNIS_HIP_LIB=$R/tennis_amd/lib/libtennis_$l.so
  rm -rf $R/gpurun_out/ab_$l
  TN_NO_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$l -- python $R/bench.py --no-cpu-baseline --single-region --steps 5 --warmup 2 > $R/gpurun_out/ab_$l.log 2>&1
  f=$(find $R/gpurun_out/ab_$l -name "*kernel_stats.csv" | head -1)
  echo "lib=$l $(grep -E "$2" $f | awk -F'","|",|,"' '{printf "%s calls=%s avg_us=%.1f  ", substr($1,2,40), $2, $4/1000}')"
  rm -rf $R/gpurun_out/ab_$l
done
