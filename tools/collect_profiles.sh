#!/bin/bash
# Local: copy the summaries of gpurun_out/$TAG into profiles/ (tracked).
TAG=${1:-r04}
O=gpurun_out/$TAG
for f in bench.json configs.jsonl callers.jsonl cfft_pmc.json bench_kernel_summary.txt configs_kernel_summary.txt bench_kernel_stats.csv sq_counters.json fri_sizes.log bench_gloo2_rehearsal.json bench_rccl_world1.json lib_sha16.txt cfft_sweep.txt quot_k.txt h2d_rate.txt; do
  [ $(wc -c < $O/$f 2>/dev/null || echo 0) -gt 10 ] && cp $O/$f profiles/${TAG}_$f
done
ls -la profiles
