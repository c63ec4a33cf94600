#!/bin/bash
# scripts/scratch/build_exp.sh <name> <extra hipcc flags...>: a copy of libtennis_hip.so with dense_strip.hip rebuilt under the flags
# (timing experiments: python scripts/kbench.py --lib scripts/scratch/libs/<name>.so)
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function "$@" -c tennis_amd/csrc/dense_strip.hip -o /tmp/ds_$name.o
objs=$(ls tennis_amd/csrc/*.o | grep -v dense_strip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/ds_$name.o -o scripts/scratch/libs/$name.so
echo built scripts/scratch/libs/$name.so
