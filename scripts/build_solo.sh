#!/bin/bash
# development aid (GPU box): SOLO durations of the header-build kernels of the staged pass (under --pmc kernels run one at a time)
#   scripts/build_solo.sh [env assignments...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
OUT="$ROOT/gpurun_out/bsolo"; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-include-regex 'k_pack|k_pair_link|k_expand_seq2|k_widen8|k_fold_complex|k_phase' --output-format csv --kernel-trace --pmc SQ_WAVES GRBM_GUI_ACTIVE -d "$OUT/a" -o run -- python3 "$ROOT/bench.py" --no-cpu --feed-dnms 0 --no-config5 --steps 2 --warmup 1 > "$OUT/a.log" 2>&1
cd "$ROOT"
python3 - <<'P'
import pandas as pd, glob
f = glob.glob('gpurun_out/bsolo/a/**/*kernel_trace.csv', recursive=True)[0]
k = pd.read_csv(f)
k['dur'] = (k.End_Timestamp - k.Start_Timestamp) / 1e3
k['nm'] = k.Kernel_Name.str.replace('void ', '').str.replace('(anonymous namespace)::', '', regex=False).str.split('(').str[0].str[:36]
print(k.groupby('nm').dur.agg(['count', 'median', 'min', 'max', 'sum']).to_string())
P
rm -rf "$OUT"
