#!/bin/bash
# development aid (GPU box): header build of an asynchronous upload on its own stream (default) against the build at first use (UZ_BUILD_LAZY=1)
run() {
  python3 bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ms', d['ms_per_step'], 'value', d['value'], 'resident ms', d['ms_per_step_resident'], 'mismatches', d['link']['result_mismatches_vs_resident'])"
}
for lazy in 0 1; do
  if [ $lazy = 1 ]; then export UZ_BUILD_LAZY=1; echo "header build at first use (compute stream):"; else unset UZ_BUILD_LAZY; echo "header build on its own stream:"; fi
  echo " config 5"; run --workload cnv
  echo " 100 k DNMs"; run
  echo " 12.5 k DNMs"; run --dnms 12500
done
