#!/bin/bash
# SQ-level PMC passes over one bench run (on the GPU box); per-kernel averages printed by scripts/pmc_sq_fmt.py
set -u
R=$PWD; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-exact-line --no-parity-live --single-region --plain-rounding --steps 2 --warmup 1"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_sq_$i
  TN_NO_SPLIT=1 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$i -- $B > $R/gpurun_out/pmc_sq_$i.log 2>&1
  find $R/gpurun_out/pmc_sq_$i -name "*kernel_trace.csv" -delete
done
cd $R
python scripts/pmc_sq_fmt.py
