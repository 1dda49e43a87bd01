# development aid: the two-build k_phase with 128-lane workgroups (build_variants/libunfazed_hip_nt128.so) x workgroups per CU x LDS arena (KiB)
TAG=$1
mkdir -p gpurun_out/$TAG
export UZ_HIP_LIB=$PWD/build_variants/libunfazed_hip_nt128.so
for cfg in "10 15" "9 17" "8 19" "12 12" "7 21" "11 13" "14 10"; do
  set -- $cfg
  echo "== lanes 128 WGS_PER_CU=$1 LDS_KB=$2" >> gpurun_out/$TAG/sweep.txt
  UZ_PHASE_WGS_PER_CU=$1 UZ_PHASE_LDS_KB=$2 timeout 300 python bench.py --no-staged --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels_ms_per_step']['phase'], d['calls']['correct_vs_truth'], d['calls']['dnms_redone_by_hbm_build_of_k_phase'])" >> gpurun_out/$TAG/sweep.txt 2>&1
done
cat gpurun_out/$TAG/sweep.txt
