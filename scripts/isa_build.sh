#!/bin/bash
# scripts/isa_build.sh <file.hip> : compiles one HIP source with -save-temps into /tmp/isa/ and prints the resource usage;
# the ISA lands in /tmp/isa/<name>-hip-amdgcn-amd-amdhsa-gfx950.s (for scripts/isa_stats.py)
set -e
mkdir -p /tmp/isa
src=$(readlink -f "$1"); shift
cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function "$@" -c "$src" -o /tmp/isa/out.o -save-temps=obj \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning: |Function Name|VGPRs:|AGPRs|VGPRs Spill|ScratchSize" | sed -e 's/.*remark: [^ ]* *//' -e 's/\[-Rpass.*//' | paste - - - - - | awk '{print $3, $4,$5,$6,$7,$8,$9,$10,$11,$12,$13,$14}'
