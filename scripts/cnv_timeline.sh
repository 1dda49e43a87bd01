#!/bin/bash
# development aid (GPU box): the kernel timeline of ONE staged step of config 5 with the host's HIP calls beside it -- what runs
# when, where the device idles, and what the host is doing meanwhile
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/cnvtl"
rocprofv3 --output-format csv --kernel-trace --hip-runtime-trace -d $ROOT/gpurun_out/cnvtl -o run -- python3 $ROOT/bench.py --workload cnv --no-cpu --steps 3 --warmup 2 ${CNV_ARGS} > /dev/null 2>&1
cd $ROOT
ls gpurun_out/cnvtl
python3 - <<'P'
import pandas as pd, numpy as np
k = pd.read_csv('gpurun_out/cnvtl/run_kernel_trace.csv').sort_values('Start_Timestamp').reset_index(drop=True)
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:30]
ph = np.nonzero(k.nm.str.startswith('k_phase<true>').values)[0]
# the staged steps come before the resident ones: the staged step's read stages are the chunked ones -- take the LAST step that has 3 read stages close together
a = pd.read_csv('gpurun_out/cnvtl/run_hip_api_trace.csv').sort_values('Start_Timestamp').reset_index(drop=True)
cn = np.nonzero(k.nm.str.contains('k_cnv_count').values)[0]
# staged steps: k_cnv_count appears 3x per step; resident 1x per step.  find the last triple whose span < 8 ms
t = k.Start_Timestamp.values
best = None
for i in range(len(cn) - 2):
    if (t[cn[i + 2]] - t[cn[i]]) / 1e6 < 6 and (best is None or cn[i] > best[0]):
        best = (cn[i], cn[i + 2])
print('last staged triple', best)
lo = t[best[0]] - 4.5e6; hi = k.End_Timestamp.values[best[1]] + 0.3e6
ks = k[(k.Start_Timestamp >= lo) & (k.End_Timestamp <= hi)].copy()
hs = a[(a.Start_Timestamp >= lo) & (a.End_Timestamp <= hi)].copy()
ks['s'] = (ks.Start_Timestamp - lo) / 1e6; ks['e'] = (ks.End_Timestamp - lo) / 1e6
hs['s'] = (hs.Start_Timestamp - lo) / 1e6; hs['e'] = (hs.End_Timestamp - lo) / 1e6
ks[['nm', 's', 'e']].to_csv('gpurun_out/cnv_step_kernels.csv', index=False, float_format='%.4f')
hs[['Function', 's', 'e']].to_csv('gpurun_out/cnv_step_hip.csv', index=False, float_format='%.4f')
P
rm -rf gpurun_out/cnvtl
