#!/bin/bash
# development aid (GPU box): what the device does during ONE resident step of the SNV bench -- every kernel and copy in order, with the gaps
# between them -- and a summary kept as profiles/r06_resident_timeline.json.   scripts/resident_timeline.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/rtl
rm -rf $OUT
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $OUT -o run -- python3 $ROOT/bench.py --no-cpu --feed-dnms 0 --no-config5 --no-staged --steps 6 --warmup 3 "$@" > $OUT.log 2>&1
cd $ROOT
python3 - <<'P'
import pandas as pd, numpy as np, json
k = pd.read_csv('gpurun_out/rtl/run_kernel_trace.csv')
m = pd.read_csv('gpurun_out/rtl/run_memory_copy_trace.csv')
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:34]
k['what'] = 'k'
m['nm'] = m.Direction.str.replace('MEMORY_COPY_', '') if 'Direction' in m.columns else 'copy'
m['what'] = 'c'
cols = ['Start_Timestamp', 'End_Timestamp', 'nm', 'what']
a = pd.concat([k[cols], m[cols]]).sort_values('Start_Timestamp').reset_index(drop=True)
# the resident steps: each holds ONE long k_phase<true> launch (> 1.5 ms); a step = from the site scan in front of it to the site scan of the next
ph = a[(a.nm.str.startswith('k_phase<true>')) & ((a.End_Timestamp - a.Start_Timestamp) > 1.5e6)]
ss = a[a.nm.str.startswith('k_site_scan')]
if len(ph) < 3:
    raise SystemExit('no resident steps found')
p1, p2 = ph.index[-2], ph.index[-1]
s1 = ss.index[ss.index < p1][-1]; s2 = ss.index[ss.index < p2][-1]
step = a.loc[s1:s2 - 1].copy()
t0 = step.Start_Timestamp.iloc[0]
step['s'] = (step.Start_Timestamp - t0) / 1e3; step['e'] = (step.End_Timestamp - t0) / 1e3
end = 0.0; rows = []; busy = 0.0; cur_s, cur_e = None, None
for _, r in step.iterrows():
    gap = r.s - end if end else 0.0
    rows.append((round(r.s, 1), round(r.e - r.s, 1), round(max(gap, 0), 1), r.what, r.nm))
    end = max(end, r.e)
total = (a.Start_Timestamp[s2] - t0) / 1e3
iv = sorted(zip(step.s, step.e)); u = 0.0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s <= ce: ce = max(ce, e)
    else: u += ce - cs; cs, ce = s, e
u += ce - cs
print('one resident step: %.1f us from site scan to site scan, device busy %.1f us, %d kernels, %d copies' % (total, u, (step.what == 'k').sum(), (step.what == 'c').sum()))
print('%9s %8s %8s  %s' % ('start', 'dur', 'gap', 'what'))
for r in rows:
    print('%9.1f %8.1f %8.1f  %s %s' % r)
gaps = sorted(rows, key=lambda r: -r[2])[:8]
json.dump({'step_us': round(total, 1), 'device_busy_us': round(u, 1), 'kernels': int((step.what == 'k').sum()), 'copies': int((step.what == 'c').sum()),
           'largest_gaps_us': [{'before': r[4], 'gap_us': r[2], 'at_us': r[0]} for r in gaps],
           'events': [{'at_us': r[0], 'dur_us': r[1], 'gap_us': r[2], 'kind': r[3], 'name': r[4]} for r in rows]},
          open('gpurun_out/resident_timeline.json', 'w'), indent=1)
P
rm -rf gpurun_out/rtl
