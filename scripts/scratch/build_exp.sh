#!/bin/bash
# scripts/scratch/build_exp.sh <name> <extra hipcc flags...>: a copy of libtennis_hip.so with dense_strip.hip (or SRC=<file>.hip) rebuilt under the flags
# (timing experiments: python scripts/kbench.py --lib scripts/scratch/libs/<name>.so)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
src=${SRC:-dense_strip}
extra=""; [ "$src" = dense_strip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function $extra "$@" -c tennis_amd/csrc/$src.hip -o /tmp/ds_$name.o
objs=$(ls tennis_amd/csrc/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ds_$name.o -o scripts/scratch/libs/$name.so
echo built scripts/scratch/libs/$name.so
