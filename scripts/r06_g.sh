export TMPDIR=/tmp; R=$PWD; cd /tmp
for lib in tennis_amd/lib/libtennis_hip.so _ab/libtennis_b14exp8.so _ab/libtennis_b14exp1.so; do
  tag=$(basename $lib .so)
  rm -rf $R/gpurun_out/pmc_b14_$tag
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_b14_$tag -- python $R/scripts/kbench.py --kernels b14 --blocks 2 --iters 3 --lib $R/$lib > $R/gpurun_out/pmc_b14_$tag.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/pmc_b14_$tag/**/*counter_collection.csv",recursive=True)
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "dense_block14" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("$tag", "dispatches", len(v), "FETCH_SIZE KB avg", sum(v)/max(1,len(v)), "=> fetch MB (x2 gfx950 correction)", 2*sum(v)/max(1,len(v))/1024)
PY
  grep b14 $R/gpurun_out/pmc_b14_$tag.log | tail -2
  find $R/gpurun_out/pmc_b14_$tag -name "*kernel_trace.csv" -delete
done
