#!/bin/bash
# A/B of environment switches on the resident pass: scripts/phase_ab.sh TAG "VAR=x VAR2=y" "" ...   ("" = defaults; UZ_HIP_LIB=... picks a build variant)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
i=0
for rep in 1 2; do
for e in "$@"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --no-staged --no-cpu --no-config5 --feed-dnms 0 --steps 10 > "$OUT/run$i.log" 2> "$OUT/run$i.err"
  grep "^{" "$OUT/run$i.log" | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('[$e]', $rep, j['ms_per_step_resident'], j['kernels_ms_per_step'], j['calls']['dnms_redone_by_hbm_build_of_k_phase'])"
  grep "phase timing" "$OUT/run$i.err" | tail -1
done; done
