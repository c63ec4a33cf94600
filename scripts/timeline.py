"""Timeline of one bench run from a rocprofv3 --kernel-trace CSV: per queue busy time and launch gaps, and how much of the wall
clock has 0 / 1 / 2 kernels running.   python scripts/timeline.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in rows))[skip:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print("kernels", len(ev), "wall %.3f ms" % ((t1 - t0) / 1e6))
byq = defaultdict(list)
for s, e, q, n in ev:
    byq[q].append((s, e, n))
for q, l in byq.items():
    busy = sum(e - s for s, e, _ in l)
    gaps = [l[i + 1][0] - l[i][1] for i in range(len(l) - 1)]
    pos = [g for g in gaps if g > 0]
    print("queue", q, "kernels", len(l), "busy %.3f ms" % (busy / 1e6), "span %.3f ms" % ((l[-1][1] - l[0][0]) / 1e6),
          "gaps>0: n=%d sum %.3f ms median %.1f us" % (len(pos), sum(pos) / 1e6, (sorted(pos)[len(pos) // 2] / 1e3 if pos else 0)))
pts = sorted([(s, 1) for s, e, _, _ in ev] + [(e, -1) for s, e, _, _ in ev])
depth, last, hist = 0, t0, defaultdict(int)
for t, d in pts:
    hist[depth] += t - last
    last = t
    depth += d
print("concurrency (fraction of wall):", {k: round(v / (t1 - t0), 3) for k, v in sorted(hist.items())})
fam = defaultdict(lambda: [0, 0])
for s, e, q, n in ev:
    k = n.split("(")[0][:60]
    fam[k][0] += e - s
    fam[k][1] += 1
for k, (d, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%-62s %8.3f ms  n=%d  avg %.1f us" % (k, d / 1e6, c, d / c / 1e3))
