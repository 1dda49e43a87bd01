#!/bin/bash
# development aid (GPU box): per-chunk kernel durations of the staged pass -- rocprofv3 kernel trace of a short bench run, medians by kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/tlk
rocprofv3 --output-format csv --kernel-trace -d $ROOT/gpurun_out/tlk -o run -- python3 $ROOT/bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 3 --warmup 1 > /dev/null 2>&1
cd $ROOT
python3 - <<'P'
import pandas as pd
k = pd.read_csv('gpurun_out/tlk/run_kernel_trace.csv')
k['dur'] = (k.End_Timestamp - k.Start_Timestamp) / 1e3
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:36]
g = k.groupby('nm').dur.agg(['count', 'median', 'sum']).sort_values('sum', ascending=False)
print(g[g['count'] >= 8].head(18).to_string())
P
rm -rf gpurun_out/tlk
