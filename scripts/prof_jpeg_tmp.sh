#!/bin/bash
R=$PWD; export TMPDIR=/tmp; cd /tmp
for v in old new; do
  rm -rf $R/gpurun_out/pj_$v
  if [ $v = old ]; then export TN_JPEG_NO_UNSTUFF=1; else unset TN_JPEG_NO_UNSTUFF; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pj_$v -- python $R/scripts/bench_jpeg.py > $R/gpurun_out/pj_$v.log 2>&1
  f=$(find $R/gpurun_out/pj_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]: print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
  find $R/gpurun_out/pj_$v -name "*kernel_trace.csv" -delete
done
