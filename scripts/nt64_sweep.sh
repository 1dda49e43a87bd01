#!/bin/bash
# One wave per DNM (WG_NT = 64 builds of k_phase) against the 256-lane build, resident pass only, over LDS arena sizes:
#   scripts/nt64_sweep.sh TAG "variant:ARENA_KB[:WGS_PER_CU]" ...     (variant "base" = the product library; ARENA_KB 0 = the host's own choice)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do
for spec in "$@"; do
  IFS=: read v kb wgs <<< "$spec"
  unset UZ_HIP_LIB UZ_PHASE_LDS_KB UZ_PHASE_WGS_PER_CU
  [ "$v" != base ] && export UZ_HIP_LIB="$ROOT/build_variants/libunfazed_hip_$v.so"
  [ -n "$kb" ] && [ "$kb" != 0 ] && export UZ_PHASE_LDS_KB=$kb
  [ -n "$wgs" ] && export UZ_PHASE_WGS_PER_CU=$wgs
  timeout 300 python bench.py --no-staged --no-cpu --no-config5 --feed-dnms 0 --steps 10 > "$OUT/$v.$kb.$wgs.$rep.log" 2> "$OUT/$v.$kb.$wgs.$rep.err"
  grep "^{" "$OUT/$v.$kb.$wgs.$rep.log" | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$spec', $rep, j['ms_per_step_resident'], j['kernels_ms_per_step'], j['calls']['dnms_redone_by_hbm_build_of_k_phase'])"
  grep "phase timing" "$OUT/$v.$kb.$wgs.$rep.err" | tail -1
done; done
