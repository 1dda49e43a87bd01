#!/bin/bash
# The files -> results pass alone (bench.py `feed`), joins on the device and on the host, for an A/B on one box.
# usage (through gpurun): scripts/feed_quick.sh TAG
TAG=${1:-feed}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for J in device host; do
  timeout 900 python bench.py --no-cpu --no-config5 --no-staged --steps 3 --warmup 1 --feed-joins $J "$@" > $OUT/bench_$J.log 2> $OUT/bench_$J.err; echo "bench $J rc $?"
  grep "^{" $OUT/bench_$J.log | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
f=j['feed']
print({k:f[k] for k in ('value_e2e','seconds_of_every_pass','result_mismatches_vs_resident','host_cpu_seconds_per_pass','joins')})
print(f['device_walk'])
print(f.get('product'))"
done
