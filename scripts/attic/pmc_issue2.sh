#!/bin/bash
# two PMC passes over the resident bench: instruction counts by class, instruction-cache behaviour.  usage: scripts/pmc_issue2.sh TAG
TAG=${1:-issue2}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --no-staged --no-cpu --steps 2 --warmup 1 $*"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES -d $OUT/i1 -o run -- python3 $ARGS > $OUT/i1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM -d $OUT/i2 -o run -- python3 $ARGS > $OUT/i2.log 2>&1
cd $ROOT
python3 scripts/pmc_summary.py $OUT/issue_summary.json $OUT/i1 $OUT/i2 > $OUT/issue_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
python3 - <<P
import json
j=json.load(open("$OUT/issue_summary.json"))
for kn in j:
    if "phase" in kn:
        print(kn)
        for k,v in sorted(j[kn]["counters_per_launch"].items()): print("  %-28s %16.0f (%d)"%(k, v["mean"], v["launches"]))
P
