#!/bin/bash
# Round-4 Merkle A/B on the GPU box: coalesced two-level subtree kernel (shipped) against round 3's lane-strided one
# (experiments build + TSTWO_MERKLE_SUBTREE_LANE_STRIDE=1): bash tools/exp_r04_merkle.sh OUT
O=$1; mkdir -p $(dirname $O); : > $O
E=$PWD/tstwo_amd/libtstwo_hip_exp.so
for rep in 1 2; do
  for args in "--cols 32 --log 22" "--cols 4 --log 24" "--cols 4 --log 23" "--cols 4 --log 20"; do
    echo "## shipped $args" >> $O; timeout -k 10 120 python tools/merkle_time.py $args --reps 200 >> $O 2>&1
    echo "## exp coalesced $args" >> $O; TSTWO_HIP_LIB=$E timeout -k 10 120 python tools/merkle_time.py $args --reps 200 >> $O 2>&1
    echo "## exp lane-stride $args" >> $O; TSTWO_HIP_LIB=$E TSTWO_MERKLE_SUBTREE_LANE_STRIDE=1 timeout -k 10 120 python tools/merkle_time.py $args --reps 200 >> $O 2>&1
  done
  echo "## bench shipped" >> $O; timeout -k 10 300 python bench.py --no-cpu --no-configs --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cfft_ms'], d['merkle_ms'])" >> $O
  echo "## bench exp lane-stride" >> $O; TSTWO_HIP_LIB=$E TSTWO_MERKLE_SUBTREE_LANE_STRIDE=1 timeout -k 10 300 python bench.py --no-cpu --no-configs --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['cfft_ms'], d['merkle_ms'])" >> $O
done
cat $O
