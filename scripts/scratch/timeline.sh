# kernel timeline of the split, pipelined bench (ON the GPU box): concurrency statistics over a steady-state window
R=$PWD; export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/tl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python $R/bench.py --no-cpu-baseline --single-region --steps 40 --warmup 10 > /dev/null 2>&1
f=$(find $R/gpurun_out/tl -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "dense_layer" in n or "conv1x1" in n or "stem_pool" in n or "head_kernel" in n:
        g = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0); w = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, g // max(w, 1), r.get("Queue_Id", "")))
ev.sort()
t0 = ev[len(ev) // 3][0]; t1 = ev[2 * len(ev) // 3][0]      # middle third = steady state
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("kernels in window", len(win), "window ms", (t1 - t0) / 1e6)
# time with k kernels in flight
pts = []
for s, e, n, wg, q in win: pts += [(s, 1), (e, -1)]
pts.sort()
cur = 0; last = t0; hist = collections.Counter()
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("fraction of time with k kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# per-kernel-family duration in this (split) run
fam = collections.defaultdict(list)
for s, e, n, wg, q in win:
    key = ("56" if "<56" in n else "28" if "<28" in n else "14" if "<14" in n else "7" if "<7" in n else "T" if "conv1x1" in n else "stem" if "stem" in n else "head", wg)
    fam[key].append((e - s) / 1e3)
for k, v in sorted(fam.items()): print(k, "n=%d avg %.1f us min %.1f max %.1f" % (len(v), sum(v) / len(v), min(v), max(v)))
P
rm -rf $R/gpurun_out/tl
