# development aid: the two-build k_phase at 5 / 6 / 7 / 8 workgroups per CU (register-capped builds in build_variants/) x LDS arena (KiB)
TAG=$1
mkdir -p gpurun_out/$TAG
for cfg in "7 7 21" "7 7 20" "6 6 23" "6 6 25" "6 6 24" "5 5 30" "5 5 29" "7 7 21"; do
  set -- $cfg
  if [ $1 != 5 ]; then export UZ_HIP_LIB=$PWD/build_variants/libunfazed_hip_w$1.so; else unset UZ_HIP_LIB; fi
  echo "== build w$1 WGS_PER_CU=$2 LDS_KB=$3" >> gpurun_out/$TAG/sweep.txt
  UZ_PHASE_WGS_PER_CU=$2 UZ_PHASE_LDS_KB=$3 timeout 300 python bench.py --no-staged --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels_ms_per_step']['phase'], d['calls']['correct_vs_truth'], d['calls']['dnms_redone_by_hbm_build_of_k_phase'])" >> gpurun_out/$TAG/sweep.txt 2>&1
done
cat gpurun_out/$TAG/sweep.txt
