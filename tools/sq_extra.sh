#!/bin/bash
# Extra SQ counter passes over tools/pmc_target.py (run on the GPU box: gpurun -- 'bash tools/sq_extra.sh TAG').
# Pass 1: activity of the two VALU ports and of the other issue types; pass 2: LDS; pass 3: vector memory.
set -o pipefail
TAG=${1:-r03x}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
T="timeout -k 10"
cd /tmp
$T 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --output-format csv -d $O/p1 -o a -- python3 $R/tools/pmc_target.py > $O/p1.log 2>&1
echo "pass 1 done"
$T 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS --output-format csv -d $O/p2 -o b -- python3 $R/tools/pmc_target.py > $O/p2.log 2>&1
echo "pass 2 done"
$T 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p3 -o c -- python3 $R/tools/pmc_target.py > $O/p3.log 2>&1
echo "pass 3 done"
$T 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES SQ_LEVEL_WAVES --output-format csv -d $O/p4 -o d -- python3 $R/tools/pmc_target.py > $O/p4.log 2>&1
echo "pass 4 done"
cd $R
python3 tools/sq_summary.py $O/p1 $O/p2 $O/p3 $O/p4 > $O/sq_extra.json
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
ls -la $O
