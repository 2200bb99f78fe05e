#!/bin/bash
# A/B timing of CFFT library builds on the GPU box: bash tools/ab_cfft.sh OUT lib1.so lib2.so ...  (paths relative to the repo)
O=$1; shift
mkdir -p $(dirname $O); : > $O
for lib in "$@"; do
  for args in "--cols 256" "--cols 256 --inv" "--cols 32" "--cols 1 --log 20"; do
    TSTWO_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/cfft_time.py $args --reps 40 >> $O 2>&1 || echo "FAILED $lib $args" >> $O
  done
done
cat $O
