#!/bin/bash
# One quick round on the GPU box: the parity tests, then the resident + staged bench without the CPU legs.
# usage (through gpurun): scripts/gpu_quick.sh TAG [pytest args]
TAG=${1:-quick}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q "$@" > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py --no-cpu --no-config5 --feed-dnms 0 --steps 10 > $OUT/bench.log 2> $OUT/bench.err; echo "bench rc $?"
grep "^{" $OUT/bench.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print({k:j[k] for k in ('value','ms_per_step','value_resident','ms_per_step_resident','kernels_ms_per_step')})
print(j.get('kernels_ms_per_step_staged')); print(j.get('link')); print(j['calls'])"
