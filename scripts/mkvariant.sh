#!/bin/bash
# scripts/mkvariant.sh NAME FILE.hip "-DFLAGS": builds _ab/libtennis_NAME.so = the in-tree objects with FILE recompiled under FLAGS
set -e
name=$1; src=$2; flags=$3
obj=_ab/${name}_$(basename ${src%.hip}).o
/opt/rocm/bin/hipcc $flags --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function -c $src -o $obj
objs=$(ls tennis_amd/csrc/*.o | grep -v "/$(basename ${src%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o _ab/libtennis_$name.so
echo _ab/libtennis_$name.so
