# development aid: k_phase at higher occupancy targets (register-capped builds in build_variants/) x LDS arena sizes
mkdir -p gpurun_out/r2j
for cfg in "6 6 20" "6 6 18" "7 7 19" "7 7 17" "8 8 16" "8 8 14" "8 7 18"; do
  set -- $cfg
  export UZ_HIP_LIB=$PWD/build_variants/libunfazed_hip_w$1.so
  echo "== build w$1 WGS_PER_CU=$2 LDS_KB=$3" >> gpurun_out/r2j/sweep.txt
  UZ_PHASE_WGS_PER_CU=$2 UZ_PHASE_LDS_KB=$3 timeout 300 python bench.py --no-staged --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels_ms_per_step']['phase'], d['calls']['correct_vs_truth'])" >> gpurun_out/r2j/sweep.txt 2>&1
done
cat gpurun_out/r2j/sweep.txt
