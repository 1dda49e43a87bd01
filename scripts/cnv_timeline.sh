#!/bin/bash
# development aid (GPU box): the kernel timeline of ONE staged step of config 5 -- what runs when, and where the device idles
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cnvtl
rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/cnvtl -o run -- python3 $GRAFT_REPO_ROOT/bench.py --workload cnv --no-cpu --steps 3 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import pandas as pd
k = pd.read_csv('gpurun_out/cnvtl/run_kernel_trace.csv')
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:30]
k = k.sort_values('Start_Timestamp').reset_index(drop=True)
# the last staged step: walk back from the last kernel to a gap of > 2 ms
t_end = k.End_Timestamp.iloc[-1]
# find steps: gaps > 1.5 ms between consecutive kernels
gaps = (k.Start_Timestamp.values[1:] - k.End_Timestamp.cummax().values[:-1]) / 1e6
import numpy as np
cut = np.nonzero(gaps > 1.0)[0]
print('kernels', len(k), 'big gaps at', cut[-12:], [round(float(g), 2) for g in gaps[cut[-12:]]])
m = pd.read_csv('gpurun_out/cnvtl/run_memory_copy_trace.csv')
print(m.columns.tolist())
k[['nm', 'Start_Timestamp', 'End_Timestamp', 'Stream_Id' if 'Stream_Id' in k.columns else 'Queue_Id']].tail(700).to_csv('gpurun_out/cnv_kernels_tail.csv', index=False)
m.tail(400).to_csv('gpurun_out/cnv_copies_tail.csv', index=False)
P
rm -rf gpurun_out/cnvtl
