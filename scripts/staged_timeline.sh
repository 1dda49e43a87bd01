#!/bin/bash
# development aid (GPU box): what the device does during ONE staged step of the SNV bench -- kernels per queue and copies, merged into busy
# spans, with the gaps between them.   scripts/staged_timeline.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/stl
rm -rf $OUT
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT -o run -- python3 $ROOT/bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 4 --warmup 2 "$@" > $OUT.log 2>&1
cd $ROOT
python3 - <<'P'
import pandas as pd, numpy as np, json
k = pd.read_csv('gpurun_out/stl/run_kernel_trace.csv').sort_values('Start_Timestamp').reset_index(drop=True)
m = pd.read_csv('gpurun_out/stl/run_memory_copy_trace.csv').sort_values('Start_Timestamp').reset_index(drop=True)
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:28]
print(list(k.columns)); print(list(m.columns))
# staged steps hold 8 k_phase<true> launches ~1.5 ms apart; resident steps one long one.  Take the last run of 8 short ones.
ph = k[k.nm.str.startswith('k_phase<true>')]
d = (ph.End_Timestamp - ph.Start_Timestamp).values / 1e6
idx = ph.index.values
short = [i for i, x in zip(idx, d) if x < 1.5]
import os
K = int(os.environ.get("UZ_TL_CHUNKS", "8"))
last8 = short[-K:]
lo = k.Start_Timestamp[last8[0]] - float(os.environ.get("UZ_TL_HEAD_MS", "3.5")) * 1e6; hi = k.End_Timestamp[last8[-1]] + 0.2e6
ks = k[(k.Start_Timestamp >= lo) & (k.End_Timestamp <= hi)].copy()
ms = m[(m.Start_Timestamp >= lo) & (m.End_Timestamp <= hi)].copy()
qcol = 'Queue_Id' if 'Queue_Id' in ks.columns else 'Stream_Id'
def spans(df, label):
    out = []
    for _, r in df.iterrows():
        s, e = (r.Start_Timestamp - lo) / 1e6, (r.End_Timestamp - lo) / 1e6
        if out and s - out[-1][1] < 0.02: out[-1][1] = max(out[-1][1], e); out[-1][2] += 1
        else: out.append([s, e, 1])
    busy = sum(e - s for s, e, _ in out)
    print('%s: busy %.2f ms in %d spans over %.2f ms' % (label, busy, len(out), (hi - lo) / 1e6))
    print('   ' + ' '.join('[%.2f-%.2f]' % (s, e) for s, e, _ in out if e - s > 0.05))
for q, g in ks.groupby(qcol):
    top = g.groupby('nm').apply(lambda x: (x.End_Timestamp - x.Start_Timestamp).sum() / 1e6).sort_values(ascending=False).head(6)
    spans(g, 'queue %s' % q)
    print('   ', {a: round(b, 2) for a, b in top.items()})
dcol = [c for c in ms.columns if 'Direction' in c or 'Kind' in c]
for key, g in ms.groupby(dcol[0] if dcol else ms.columns[0]):
    byt = g['Bytes'].sum() if 'Bytes' in g.columns else 0
    spans(g, 'copies %s (%.0f MB)' % (key, byt / 1e6))
P
rm -rf gpurun_out/stl
