#!/bin/bash
# Round-4 plan A/B, second batch: bash tools/exp_r04_cfft2.sh OUT
O=$1; mkdir -p $(dirname $O); : > $O
E=$PWD/tstwo_amd/libtstwo_hip_exp.so
t() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "## $label" >> $O
  env "${envs[@]}" timeout -k 10 180 python tools/cfft_time.py "$@" --reps 60 >> $O 2>&1 || echo "FAILED $label" >> $O; }
TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=12 TSTWO_CFFT_KA=10 TSTWO_CFFT_LOGTA=15 python tools/plan_digest.py --log 22 >> $O 2>&1
python tools/plan_digest.py --log 22 >> $O 2>&1
for rep in 1 2; do
for dir in "" "--inv"; do
  t "n22 shipped (13+9, LOGT14)" -- --cols 256 --log 22 $dir
  t "n22 13+9 LOGT15" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=15 -- --cols 256 --log 22 $dir
  t "n22 12+10 LOGT15" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=12 TSTWO_CFFT_KA=10 TSTWO_CFFT_LOGTA=15 -- --cols 256 --log 22 $dir
  t "n22 32 cols shipped" -- --cols 32 --log 22 $dir
  t "n22 32 cols 13+9 LOGT15" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=15 -- --cols 32 --log 22 $dir
done
done
cat $O
