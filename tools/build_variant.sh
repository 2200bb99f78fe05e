#!/bin/bash
# Experimental library builds for A/B timing: bash tools/build_variant.sh NAME TU [extra hipcc flags...]
# recompiles csrc/TU.hip with the extra flags and links it with the other (already built) objects into build/exp/NAME.so
set -e
NAME=$1; TU=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/exp/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $R/tstwo_amd/csrc/$TU.hip -o $R/build/exp/obj/${NAME}_$TU.o
OBJS=""
for s in context field_ops cfft fri merkle quotients comm; do
  if [ $s = $TU ]; then OBJS="$OBJS $R/build/exp/obj/${NAME}_$TU.o"; else OBJS="$OBJS $R/tstwo_amd/csrc/obj/$s.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/exp/$NAME.so $OBJS -ldl
echo built build/exp/$NAME.so
