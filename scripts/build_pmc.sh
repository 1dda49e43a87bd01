#!/bin/bash
# development aid (GPU box): the header-build kernels of the staged pass under PMC counters (kernels run one at a time under --pmc:
# their durations here are solo durations).  usage: scripts/build_pmc.sh  -> prints per-kernel averages
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/bpmc
rm -rf $OUT; mkdir -p $OUT
ONLY='--kernel-include-regex k_pack_rec|k_off_block|k_pair_link|k_expand_seq2|k_patch_exc|k_phase_bounds'
ARGS="$ROOT/bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 2 --warmup 1"
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/a -o run -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 $ONLY --output-format csv --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $OUT/b -o run -- python3 $ARGS > $OUT/b.log 2>&1
cd $ROOT
python3 scripts/pmc_rows.py $OUT/a
python3 scripts/pmc_rows.py $OUT/b
python3 - <<'P'
import pandas as pd, glob
f = glob.glob('gpurun_out/bpmc/a/**/*kernel_trace.csv', recursive=True)[0]
k = pd.read_csv(f)
k['dur'] = (k.End_Timestamp - k.Start_Timestamp) / 1e3
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:36]
print(k.groupby('nm').dur.agg(['count', 'median', 'min', 'max', 'sum']).to_string())
P
find $OUT -name "*.csv" -size +2M -delete
