#!/bin/bash
# development aid (GPU box): config 5's staged step over chunk plans
for v in "3 1.0" "3 0.5" "3 0.3" "4 0.5" "4 0.3" "5 0.3"; do
  set -- $v
  python3 bench.py --workload cnv --no-cpu --steps 10 --warmup 3 --chunks $1 --first-chunk $2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunks $1 first $2: ms', d['ms_per_step'], 'value', d['value'], 'mismatch', d['link']['result_mismatches_vs_resident'])"
done
