#!/bin/bash
# Experimental library builds for A/B timing: bash tools/build_variant.sh NAME TU [extra hipcc flags...]
# recompiles csrc/TU.hip with -DTSTWO_EXPERIMENTS and the extra flags and links it with the other objects of the EXPERIMENTS
# build (python -m tstwo_amd.build --experiments; the TSTWO_* switches of DESIGN.md §8 are live in it) into build/exp/NAME.so.
# Use it through TSTWO_HIP_LIB=build/exp/NAME.so.
set -e
NAME=$1; TU=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build/exp/obj
python3 -m tstwo_amd.build --experiments > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DTSTWO_EXPERIMENTS "$@" -c $R/tstwo_amd/csrc/$TU.hip -o $R/build/exp/obj/${NAME}_$TU.o
OBJS=""
for s in context field_ops cfft fri merkle quotients comm; do
  if [ $s = $TU ]; then OBJS="$OBJS $R/build/exp/obj/${NAME}_$TU.o"; else OBJS="$OBJS $R/tstwo_amd/csrc/obj/exp/$s.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/build/exp/$NAME.so $OBJS -ldl
echo built build/exp/$NAME.so
