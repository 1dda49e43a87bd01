mkdir -p gpurun_out/r2f
for cfg in "5 28" "5 24" "5 31" "4 38" "4 28" "6 24" "3 50"; do
  set -- $cfg
  echo "== WGS_PER_CU=$1 LDS_KB=$2" >> gpurun_out/r2f/sweep.txt
  UZ_PHASE_WGS_PER_CU=$1 UZ_PHASE_LDS_KB=$2 timeout 300 python bench.py --no-staged --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels_ms_per_step'], d['ms_per_step_resident'])" >> gpurun_out/r2f/sweep.txt 2>&1
done
cat gpurun_out/r2f/sweep.txt
