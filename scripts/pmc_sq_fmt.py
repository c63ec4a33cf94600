import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sorted(glob.glob("gpurun_out/pmc_sq_*")):
    if not os.path.isdir(d): continue
    fs = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in fs[-1:]:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = None
            for key in ("dense_strip_kernel<56", "dense_strip_kernel<28", "dense_block7_kernel", "dense_block14_kernel", "dense_layer_kernel<56", "dense_layer_kernel<28", "dense_layer_kernel<14", "dense_layer_kernel<7",
                        "stem_pool", "conv1x1_kernel"):
                if key in n: k = key
            if k:
                a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    base = cs.get("SQ_WAVE_CYCLES", [1, 1.0]); wc = base[1] / max(1, base[0])
    for c, (n, v) in cs.items():
        print("   %-28s %14.0f  (%.3f of SQ_WAVE_CYCLES)" % (c, v / n, v / n / wc if wc else 0))
