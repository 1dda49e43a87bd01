#!/bin/bash
# development aid (GPU box): ONE pass of the feed leg as busy spans -- inflate alone, every kernel, the copies -- and what the device waits for.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/ftl
rm -rf $OUT
UZ_BENCH_NO_PRODUCT=1 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT -o run -- python3 $ROOT/bench.py --no-cpu --no-config5 --no-staged --steps 1 --warmup 0 --feed-reps 2 "$@" > $OUT.log 2>&1
cd $ROOT
python3 - <<'P'
import pandas as pd, numpy as np
k = pd.read_csv('gpurun_out/ftl/run_kernel_trace.csv').sort_values('Start_Timestamp').reset_index(drop=True)
m = pd.read_csv('gpurun_out/ftl/run_memory_copy_trace.csv').sort_values('Start_Timestamp').reset_index(drop=True)
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:40]
inf = k[k.nm.str.startswith('k_bgzf_inflate')]
# passes: gaps of more than 30 ms between inflate launches separate them
st = inf.Start_Timestamp.values
cut = [0] + [i for i in range(1, len(st)) if st[i] - inf.End_Timestamp.values[i - 1] > 30e6] + [len(st)]
a, b = cut[-2], cut[-1]
lo, hi = st[a] - 2e6, inf.End_Timestamp.values[b - 1] + 25e6
ks = k[(k.Start_Timestamp >= lo) & (k.Start_Timestamp <= hi)]
ms = m[(m.Start_Timestamp >= lo) & (m.Start_Timestamp <= hi)]
def union(df):
    iv = sorted(zip(df.Start_Timestamp.values, df.End_Timestamp.values))
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot / 1e6
print('last pass: %d inflate launches, window %.1f ms' % (b - a, (hi - lo) / 1e6))
print('busy ms: inflate %.1f | every kernel %.1f | kernels but inflate %.1f | copies %.1f' % (
    union(ks[ks.nm.str.startswith('k_bgzf_inflate')]), union(ks), union(ks[~ks.nm.str.startswith('k_bgzf_inflate')]), union(ms)))
g = ks.assign(dur=(ks.End_Timestamp - ks.Start_Timestamp) / 1e3).groupby('nm').dur.agg(['count', 'median', 'sum']).sort_values('sum', ascending=False)
print(g.head(30).to_string())
P
grep "^{" gpurun_out/ftl.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['feed']['value_e2e'], j['feed']['seconds_of_every_pass'])"
rm -rf gpurun_out/ftl
