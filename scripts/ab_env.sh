#!/bin/bash
# A/B of environment switches on one box (staged bench): scripts/ab_env.sh "VAR1=x VAR2=y" "VAR3=z" ...  ("" = defaults)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for rep in 1 2; do
for e in "$@"; do
  env $e python bench.py --no-cpu --no-config5 --feed-dnms 0 --steps 10 ${UZ_AB_ARGS} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e]', $rep, 'staged', j['ms_per_step'], 'resident', j['ms_per_step_resident'], 'mism', j['link']['result_mismatches_vs_resident'])"
done; done
