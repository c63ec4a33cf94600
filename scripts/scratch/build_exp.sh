#!/bin/bash
# scripts/scratch/build_exp.sh <name> <extra hipcc flags...>: a copy of libtennis_hip.so with dense_strip_w56.hip (or SRC=<file without .hip>) rebuilt under the flags
# (timing experiments: python scripts/kbench.py --lib scripts/scratch/libs/<name>.so)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
src=${SRC:-dense_strip_w56}
extra=""; case "$src" in dense_strip_w*) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function $extra "$@" -c tennis_amd/csrc/$src.hip -o /tmp/ds_$name.o
objs=$(ls tennis_amd/csrc/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ds_$name.o -o scripts/scratch/libs/$name.so
echo built scripts/scratch/libs/$name.so
