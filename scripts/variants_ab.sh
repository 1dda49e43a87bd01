#!/bin/bash
# A/B of kernel build variants on one box (resident pass only): scripts/variants_ab.sh TAG name1 name2 ...  (build_variants/libunfazed_hip_NAME.so; "base" = the product library)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset UZ_HIP_LIB; else export UZ_HIP_LIB=$ROOT/build_variants/libunfazed_hip_$v.so; fi
  python bench.py --no-staged --no-cpu --no-config5 --feed-dnms 0 --steps 10 > $OUT/$v.$rep.log 2> $OUT/$v.$rep.err
  grep "^{" $OUT/$v.$rep.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v', $rep, j['ms_per_step_resident'], j['kernels_ms_per_step'], j['calls']['dnms_redone_by_hbm_build_of_k_phase'])"
  grep "phase timing" $OUT/$v.$rep.err | tail -1
done; done
