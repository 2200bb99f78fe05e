#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh TAG'): regenerates every artefact profiles/ holds.
# Output goes to gpurun_out/$TAG/ (scratch); tools/collect_profiles.sh copies the summaries into profiles/.
# Every profiled command sits behind `timeout -k`, and a progress line is printed after each stage.
set -o pipefail
TAG=${1:-r04}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
T="timeout -k 10"
cd /tmp
# 1. PMC passes (separate runs, --pmc only): HBM traffic with the guide's gfx950 corrections, then SQ counters
$T 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/pmc_target.py > $O/pmc_fetch.log 2>&1
$T 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/pmc_target.py > $O/pmc_write.log 2>&1
echo "pmc traffic done"
$T 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $O/pmc_sq1 -o s1 -- python3 $R/tools/pmc_target.py > $O/pmc_sq1.log 2>&1
$T 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq2 -o s2 -- python3 $R/tools/pmc_target.py > $O/pmc_sq2.log 2>&1
echo "pmc sq done"
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 22 256 > $O/cfft_pmc.json
python3 tools/sq_summary.py $O/pmc_sq1 $O/pmc_sq2 > $O/sq_counters.json
# 2. the bench line (configs 1-4 inside), with the traffic of THIS build
# (bench.py collects roofline.traffic itself in child rocprofv3 --pmc passes; cfft_pmc.json above is the per-kernel record)
$T 800 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench done"; tail -c 300 $O/bench.json
# 3. kernel trace + stats of the same command (without the CPU leg and the config sweep) and of the config sweep
cd /tmp
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu --no-configs --no-pmc > $O/stats.log 2>&1
echo "stats done"
$T 300 rocprofv3 --kernel-trace --output-format csv -d $O/stats_configs -o configs -- python3 $R/tools/bench_configs.py --no-cpu > $O/stats_configs.log 2>&1
echo "stats configs done"
cd $R
python3 tools/prof_summary.py $O/stats > $O/bench_kernel_summary.txt
python3 tools/prof_summary.py $O/stats_configs > $O/configs_kernel_summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
$T 300 python3 tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
echo "configs done"
$T 300 python3 tools/bench_callers.py > $O/callers.jsonl 2> $O/callers.err
echo "callers done"
$T 300 python3 tools/bench_fri_sizes.py > $O/fri_sizes.log 2>&1
echo "fri sizes done"
# the N = 2 control flow of the strong-scaling bench on the one GPU (roots through the host) and the RCCL path at world size 1
# (round 4: bench.py --gpus 2 starts its own two ranks; both share the one GPU here, so the roots travel through the host: gloo)
TSTWO_BENCH_COLLECTIVE=gloo $T 400 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-configs > $O/bench_gloo2_rehearsal.json 2> $O/bench_gloo2.err
TSTWO_FORCE_DIST=1 $T 300 python3 bench.py --steps 5 --warmup 2 --no-configs --no-cpu --no-pmc > $O/bench_rccl_world1.json 2> $O/bench_rccl1.err
echo "rehearsals done"
# round 4: size sweep of the transform, k sample batches over one column list, host -> device rates
$T 600 bash tools/cfft_sweep.sh > $O/cfft_sweep.txt 2>&1
($T 200 python3 tools/quot_k_time.py --kmax 5; $T 200 python3 tools/quot_k_time.py --kmax 4 --cols 256 --log 20) > $O/quot_k.txt 2>&1
$T 200 python3 tools/h2d_rate.py > $O/h2d_rate.txt 2>&1
echo "sweeps done"
rm -rf $O/stats $O/stats_configs $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2      # raw traces are large; summaries are what is kept
ls -la $O
sha256sum $R/tstwo_amd/libtstwo_hip.so | cut -c1-16 > $O/lib_sha16.txt
