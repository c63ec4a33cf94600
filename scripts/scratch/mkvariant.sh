#!/bin/bash
# scripts/scratch/mkvariant.sh NAME FILE.hip "-DFOO=1 ..." : libtennis_NAME.so = the current objects with FILE rebuilt under the given defines
set -e
N=$1; F=$2; D=$3
C=tennis_amd/csrc
/opt/rocm/bin/hipcc $D --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c $C/$F.hip -o /tmp/$F.$N.o
OBJS=$(ls $C/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/$F.$N.o -o tennis_amd/lib/libtennis_$N.so
echo built tennis_amd/lib/libtennis_$N.so
