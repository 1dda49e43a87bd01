# development aid: k_phase with narrower workgroups (build_variants/libunfazed_hip_nt{64,128}.so) x workgroups per CU x LDS arena
mkdir -p gpurun_out/r2o
for cfg in "128 10 12" "128 10 13" "128 8 16" "64 20 5" "64 16 7" "64 20 6" "64 12 10"; do
  set -- $cfg
  export UZ_HIP_LIB=$PWD/build_variants/libunfazed_hip_nt$1.so
  echo "== lanes $1 WGS_PER_CU=$2 LDS_KB=$3" >> gpurun_out/r2o/sweep.txt
  UZ_PHASE_WGS_PER_CU=$2 UZ_PHASE_LDS_KB=$3 timeout 200 python bench.py --no-staged --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels_ms_per_step']['phase'], d['calls']['correct_vs_truth'])" >> gpurun_out/r2o/sweep.txt 2>&1
done
cat gpurun_out/r2o/sweep.txt
