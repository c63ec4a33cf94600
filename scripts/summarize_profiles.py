"""Turn rocprofv3 CSV output (gpurun_out/prof_*) into the summaries committed under profiles/.

usage: python scripts/summarize_profiles.py <tag>        e.g.  r01_d
  gpurun_out/prof_stats/**/*kernel_stats.csv      -> profiles/<tag>_kernel_stats_bench_steps100.csv
  gpurun_out/prof_fetch|prof_write/**/*counter_collection.csv -> profiles/<tag>_pmc_traffic.json
PMC units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE
are KB per dispatch; FETCH_SIZE counts 64 B per 128-B request -> hbm_bytes = (2*FETCH + WRITE) * 1024.
"""
import csv, glob, json, os, shutil, sys, collections


def newest(pattern):
    """gpurun_out/ accumulates the files of earlier calls: use only the newest match."""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1:]

tag = sys.argv[1]
st = newest("gpurun_out/prof_stats/**/*kernel_stats.csv")
if st:
    shutil.copy(st[0], f"profiles/{tag}_kernel_stats_bench_steps100.csv")


def family(name):
    for key in ("dense_block7_kernel", "dense_block14_kernel", "dense_strip_kernel<56", "dense_strip_kernel<28", "dense_layer_kernel<56", "dense_layer_kernel<28", "dense_layer_kernel<14", "dense_layer_kernel<7",
                "dense_block", "stem_pool_kernel", "trans_ws_kernel", "conv1x1_kernel", "head_kernel", "stem_kernel", "maxpool_kernel"):
        if key in name:
            return key
    return None


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in newest(f"gpurun_out/{d}/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            fam = family(r["Kernel_Name"])
            if fam:
                acc[fam][0] += 1
                acc[fam][1] += float(r["Counter_Value"])
    return acc


def collect_raw(d, counter):
    rows = []
    for f in newest(f"gpurun_out/{d}/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


fe, wr = collect("prof_fetch", "FETCH_SIZE"), collect("prof_write", "WRITE_SIZE")
if fe and wr:
    out = {"command": "rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv -- python bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline (one pass per counter)",
           "units": "FETCH_SIZE / WRITE_SIZE are KB per dispatch as reported by rocprofv3; gfx950 correction "
                    "(MI355X_MICROARCH.md, HBM): hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
           "families": {}}
    for fam in fe:
        n, f = fe[fam]
        w = wr[fam][1] / max(1, wr[fam][0])
        f /= n
        out["families"][fam] = {"dispatches": n, "FETCH_SIZE_KB_per_dispatch": round(f, 1),
                                "WRITE_SIZE_KB_per_dispatch": round(w, 1),
                                "hbm_bytes_per_dispatch_corrected": int((2 * f + w) * 1024)}
    # bench.py's kernel families: per-layer launches (56^2, 28^2 blocks) and chained whole-block launches
    raw_f, raw_w = collect_raw("prof_fetch", "FETCH_SIZE"), collect_raw("prof_write", "WRITE_SIZE")
    for fam, pred in (("dense_layer_fused", lambda n: "dense_layer_kernel" in n and "false>" in n),
                      ("dense_block_chained", lambda n: "dense_layer_kernel" in n and "true>" in n)):
        fv = [v for n, v in raw_f if pred(n)]
        wv = [v for n, v in raw_w if pred(n)]
        if fv and wv:
            f, w = sum(fv) / len(fv), sum(wv) / len(wv)
            out["families"][fam] = {"dispatches": len(fv), "FETCH_SIZE_KB_per_dispatch": round(f, 1),
                                    "WRITE_SIZE_KB_per_dispatch": round(w, 1),
                                    "hbm_bytes_per_dispatch_corrected": int((2 * f + w) * 1024)}
    json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out["families"], indent=1))
