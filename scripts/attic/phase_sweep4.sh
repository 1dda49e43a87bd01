# development aid: LDS arena size of k_phase<true> (KiB; workgroups per CU follow from it) -> kernel time, DNMs redone by the HBM build
mkdir -p gpurun_out/$1
for kb in 28 32 36 40 48 52; do
  echo "== LDS_KB=$kb" >> gpurun_out/$1/sweep.txt
  UZ_PHASE_LDS_KB=$kb timeout 200 python bench.py --no-staged --no-cpu --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels_ms_per_step']['phase'], d['calls']['correct_vs_truth'], d['calls']['dnms_redone_by_hbm_build_of_k_phase'])" >> gpurun_out/$1/sweep.txt 2>&1
done
cat gpurun_out/$1/sweep.txt
