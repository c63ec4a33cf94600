#!/bin/bash
# SQ-level PMC passes over the three transition launches of scripts/kbench.py (on the GPU box)
set -u
R=$PWD; export TMPDIR=/tmp
B="python $R/scripts/kbench.py --kernels tr --iters 4"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_tr_$i
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tr_$i -- $B > $R/gpurun_out/pmc_tr_$i.log 2>&1
  find $R/gpurun_out/pmc_tr_$i -name "*kernel_trace.csv" -delete
done
cd $R
python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sorted(glob.glob("gpurun_out/pmc_tr_*")):
    if not os.path.isdir(d): continue
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "conv1x1_kernel" not in n: continue
            k = n[n.index("conv1x1_kernel"):][:60] + " grid " + r.get("Grid_Size", "?")
            a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in acc.items():
    print(k)
    base = cs.get("SQ_WAVE_CYCLES", [1, 1.0]); wc = base[1] / max(1, base[0])
    for c, (n, v) in cs.items():
        print("   %-28s %16.0f  (%.3f of SQ_WAVE_CYCLES)" % (c, v / n, v / n / wc if wc else 0))
PY
