#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh TAG'): regenerates every artefact profiles/ holds.
# Output goes to gpurun_out/$TAG/ (scratch); tools/collect_profiles.sh copies the summaries into profiles/.
set -eo pipefail
TAG=${1:-r01}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo "bench done"; tail -c 400 $O/bench.json
python3 tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
echo "configs done"
python3 tools/bench_callers.py > $O/callers.jsonl 2> $O/callers.err
echo "callers done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu > $O/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --output-format csv -d $O/stats_configs -o configs -- python3 $R/tools/bench_configs.py > $O/stats_configs.log 2>&1
echo "stats configs done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/pmc_target.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/pmc_target.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $O/pmc_sq1 -o s1 -- python3 $R/tools/pmc_target.py > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2 -o s2 -- python3 $R/tools/pmc_target.py > $O/pmc_sq2.log 2>&1
echo "pmc done"
cd $R
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/cfft_pmc.json
python3 tools/sq_summary.py $O/pmc_sq1 $O/pmc_sq2 > $O/sq_counters.json
python3 tools/prof_summary.py $O/stats > $O/bench_kernel_summary.txt
python3 tools/prof_summary.py $O/stats_configs > $O/configs_kernel_summary.txt
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/stats $O/stats_configs $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2      # raw traces are large; summaries are what is kept
ls -la $O
