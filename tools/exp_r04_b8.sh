#!/bin/bash
# Round-4: the 8-words-per-lane bottom pass (experiments build, TSTWO_CFFT_B8=1) against the shipped 16-words-per-lane one: bash tools/exp_r04_b8.sh OUT
O=$1; mkdir -p $(dirname $O); : > $O
E=$PWD/tstwo_amd/libtstwo_hip_exp.so
for n in 13 16 20 22; do
  python tools/plan_digest.py --log $n >> $O 2>&1
  TSTWO_HIP_LIB=$E TSTWO_CFFT_B8=1 python tools/plan_digest.py --log $n >> $O 2>&1
done
t() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "## $label" >> $O
  env "${envs[@]}" timeout -k 10 180 python tools/cfft_time.py "$@" >> $O 2>&1 || echo "FAILED $label" >> $O; }
for rep in 1 2; do
for dir in "" "--inv"; do
  t "n22 256 cols shipped" -- --cols 256 --log 22 --reps 60 $dir
  t "n22 256 cols B8" TSTWO_HIP_LIB=$E TSTWO_CFFT_B8=1 -- --cols 256 --log 22 --reps 60 $dir
done
done
t "n20 1 col shipped" -- --cols 1 --log 20 --reps 300
t "n20 1 col B8 (13 + 7)" TSTWO_HIP_LIB=$E TSTWO_CFFT_B8=1 TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 -- --cols 1 --log 20 --reps 300
t "n13 256 cols shipped" -- --cols 256 --log 13 --reps 300
t "n13 256 cols B8" TSTWO_HIP_LIB=$E TSTWO_CFFT_B8=1 -- --cols 256 --log 13 --reps 300
cat $O
