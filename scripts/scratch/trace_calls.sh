# per-call durations of the kernels matching $1 in one bench run (ON the GPU box)
R=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/trc
env $EXTRA_ENV TN_NO_SPLIT=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trc -- python $R/bench.py --no-cpu-baseline --single-region --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find $R/gpurun_out/trc -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$1" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
seq = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size")) for r in rows if pat in r["Kernel_Name"]]
d = collections.OrderedDict()
for n, dt, g, w in seq: d.setdefault((n[:70], g, w), []).append(dt)
for k, v in d.items(): print(k, "n=%d" % len(v), "avg %.1f us" % (sum(v) / len(v) / 1e3), "min %.1f" % (min(v) / 1e3))
P
rm -rf $R/gpurun_out/trc
