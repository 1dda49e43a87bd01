#!/bin/bash
# Instruction-issue picture of k_phase (run through gpurun): instruction counts by class, busy cycles by class, lane utilisation.
# usage: scripts/pmc_issue.sh TAG [extra bench args]  -> gpurun_out/TAG/issue_summary.json
TAG=${1:-issue}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --no-staged --no-cpu --steps 2 --warmup 1 $*"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_INSTS_VMEM SQ_WAVES -d $OUT/i1 -o run -- python3 $ARGS > $OUT/i1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU -d $OUT/i2 -o run -- python3 $ARGS > $OUT/i2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT_NO_LDS SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/i3 -o run -- python3 $ARGS > $OUT/i3.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/i4 -o run -- python3 $ARGS > $OUT/i4.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum -d $OUT/i5 -o run -- python3 $ARGS > $OUT/i5.log 2>&1
cd $ROOT
python3 scripts/pmc_summary.py $OUT/issue_summary.json $OUT/i1 $OUT/i2 $OUT/i3 $OUT/i4 $OUT/i5 > $OUT/issue_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
du -sh $OUT
