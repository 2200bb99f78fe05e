#!/bin/bash
# Round-4 CFFT plan experiments on the GPU box: bash tools/exp_r04_cfft.sh OUT
# (experiments build = tstwo_amd/libtstwo_hip_exp.so: TSTWO_CFFT_KB / KA / LOGTA live; build/exp/a_sb.so = scalar-base addressing in every strided kernel)
O=$1; mkdir -p $(dirname $O); : > $O
E=$PWD/tstwo_amd/libtstwo_hip_exp.so
t() { # label, env..., -- args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "## $label" >> $O
  env "${envs[@]}" timeout -k 10 180 python tools/cfft_time.py "$@" --reps 40 >> $O 2>&1 || echo "FAILED $label" >> $O
}
for dir in "" "--inv"; do
  t "n22 shipped (13+9, LOGT14)" -- --cols 256 --log 22 $dir
  t "n22 13+9 LOGT15 (256-byte rows)" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=15 -- --cols 256 --log 22 $dir
  t "n22 14+8 LOGT15 (512-byte rows)" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=14 TSTWO_CFFT_KA=8 TSTWO_CFFT_LOGTA=15 -- --cols 256 --log 22 $dir
  t "n22 scalar-base LOGT14" TSTWO_HIP_LIB=$PWD/build/exp/a_sb.so -- --cols 256 --log 22 $dir
  t "n23 shipped (14+9, LOGT14)" -- --cols 128 --log 23 $dir
  t "n23 14+9 LOGT15" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=14 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=15 -- --cols 128 --log 23 $dir
  t "n23 13+10 LOGT15" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=10 TSTWO_CFFT_LOGTA=15 -- --cols 128 --log 23 $dir
  t "n24 shipped (14+10, LOGT15)" -- --cols 64 --log 24 $dir
  t "n24 round 3 (13+6+5)" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=14 -- --cols 64 --log 24 $dir
  t "n24 32 cols shipped" -- --cols 32 --log 24 $dir
  t "n24 32 cols round 3" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=14 -- --cols 32 --log 24 $dir
done
cat $O
