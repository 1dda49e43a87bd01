#!/bin/bash
# cProfile of the product route (files -> BED text) on the GPU box: scripts/e2e_profile.sh TAG [N] [M]
TAG=${1:-e2eprof}; N=${2:-20000}; M=${3:-20000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
python -m cProfile -o $OUT/e2e.prof scripts/e2e_time.py $N $M > $OUT/e2e.log 2>&1
tail -6 $OUT/e2e.log
python - <<PY
import pstats
p = pstats.Stats("$OUT/e2e.prof"); p.sort_stats("cumulative").print_stats(45)
PY
