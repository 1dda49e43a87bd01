#!/bin/bash
# Kernel statistics and counters of the device's BGZF inflate (GPU box, through gpurun): scripts/inflate_probe.py N under rocprofv3.
# usage: scripts/profile_inflate.sh [N=20000]   -> gpurun_out/inflate/{stats,insts,sq}/..., summary in gpurun_out/inflate/
N=${1:-20000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/inflate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ONLY='--kernel-include-regex k_bgzf_inflate'
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -o run -- python3 $ROOT/scripts/inflate_probe.py $N > $OUT/stats.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES -d $OUT/insts -o run -- python3 $ROOT/scripts/inflate_probe.py $N > $OUT/insts.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/sq -o run -- python3 $ROOT/scripts/inflate_probe.py $N > $OUT/sq.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o run -- python3 $ROOT/scripts/inflate_probe.py $N > $OUT/fetch.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o run -- python3 $ROOT/scripts/inflate_probe.py $N > $OUT/write.log 2>&1
cd $ROOT
python3 scripts/pmc_summary.py $OUT/pmc_summary.json $OUT/insts $OUT/sq $OUT/fetch $OUT/write > $OUT/pmc_summary.txt 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
grep "device\|BAM:" $OUT/stats.log > $OUT/probe.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT
