#!/bin/bash
# Round-4 strided-pass A/B: the 2^14 tile on 512 lanes x 32 words (two workgroups per CU) against the shipped 2^15 tile: bash tools/exp_r04_cfft3.sh OUT
O=$1; mkdir -p $(dirname $O); : > $O
E=$PWD/tstwo_amd/libtstwo_hip_exp.so
TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=14 TSTWO_CFFT_AV=2 python tools/plan_digest.py --log 22 >> $O 2>&1
python tools/plan_digest.py --log 22 >> $O 2>&1
t() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "## $label" >> $O
  env "${envs[@]}" timeout -k 10 180 python tools/cfft_time.py "$@" --reps 60 >> $O 2>&1 || echo "FAILED $label" >> $O; }
for rep in 1 2; do
for dir in "" "--inv"; do
  t "n22 shipped (13+9, LOGT15 V2 1024 lanes)" -- --cols 256 --log 22 $dir
  t "n22 13+9 LOGT14 V2 (512 lanes x 32 words, 2 WG/CU)" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=14 TSTWO_CFFT_AV=2 -- --cols 256 --log 22 $dir
  t "n22 13+9 LOGT14 V1 (round 3)" TSTWO_HIP_LIB=$E TSTWO_CFFT_KB=13 TSTWO_CFFT_KA=9 TSTWO_CFFT_LOGTA=14 -- --cols 256 --log 22 $dir
done
done
cat $O
