#!/bin/bash
# per-dispatch timeline of ONE tn_jpeg_decode call (256 x 720p) from a rocprofv3 kernel trace: bash scripts/jpeg_timeline.sh  (on the GPU box)
R=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/pj_tr
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pj_tr -- python $R/scripts/bench_jpeg.py --iters 3 --decoders "" > $R/gpurun_out/pj_tr.log 2>&1
f=$(find $R/gpurun_out/pj_tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
rows=[r for r in rows if 'jpeg' in r['Kernel_Name']]
names=[r['Kernel_Name'] for r in rows]
first=[i for i,n in enumerate(names) if 'unstuff_count' in n or ('sync' in n and (i==0 or 'color' in names[i-1]))]
last=first[-1]
t0=int(rows[last]['Start_Timestamp'])
for r in rows[last:]:
    n=r['Kernel_Name']; n=n[n.find('jpeg_'):][:44]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  +{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  {n}")
PY
rm -rf $R/gpurun_out/pj_tr
