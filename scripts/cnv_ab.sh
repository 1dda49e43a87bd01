cd ${GRAFT_REPO_ROOT:-/root/repo}
for e in "" "UZ_PHASE_ARENA_PERMILLE=1000" "UZ_PHASE_LDS_KB=14" "UZ_PHASE_LDS_KB=20" "UZ_PHASE_LDS_KB=30"; do
  env $e python bench.py --workload cnv --no-cpu --no-staged --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('[$e]', j['ms_per_step_resident'], j['kernels_ms_per_step']['phase'], j['calls']['dnms_redone_by_hbm_build_of_k_phase'])"
done
UZ_PHASE_SPEC_LOG=1 python bench.py --workload cnv --no-cpu --no-staged --steps 2 2>&1 | grep uz_launch_phase | tail -2
