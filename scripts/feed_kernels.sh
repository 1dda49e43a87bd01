#!/bin/bash
# development aid (GPU box): kernel durations of the feed pass (device walk) -- rocprofv3 kernel trace, medians by kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/tlf
UZ_BENCH_NO_PRODUCT=1 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $ROOT/gpurun_out/tlf -o run -- python3 $ROOT/bench.py --no-cpu --no-config5 --steps 1 --warmup 0 "$@" > $ROOT/gpurun_out/tlf.log 2>&1
cd $ROOT
python3 - <<'P'
import pandas as pd
k = pd.read_csv('gpurun_out/tlf/run_kernel_trace.csv')
k['dur'] = (k.End_Timestamp - k.Start_Timestamp) / 1e3
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:36]
g = k.groupby('nm').dur.agg(['count', 'median', 'max', 'sum']).sort_values('sum', ascending=False)
print(g[g.index.str.contains('bam|inflate|crc|desc_filter|tab_insert|pack_rec|off_block|phase|extract|walk|join|final|radix|sort|scan|inverse|need|lookback|site|window')].to_string())
try:
    m = pd.read_csv('gpurun_out/tlf/run_memory_copy_trace.csv')
    m['dur'] = (m.End_Timestamp - m.Start_Timestamp) / 1e3
    big = m[m.dur > 500]
    print(big.groupby('Direction').dur.agg(['count', 'median', 'sum']).to_string())
except Exception as e:
    print('no copy trace', e)
P
grep "^{" gpurun_out/tlf.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['feed']['value_e2e'], j['feed']['seconds'])"
rm -rf gpurun_out/tlf
