#!/bin/bash
# development aid (GPU box): config 5's staged step over chunk plans (chunks, first-chunk, last-chunk)
for v in ${PLANS:-"3 0.5 0.5" "3 0.5 0.3" "3 0.5 0.7" "3 0.5 1.0" "3 0.7 0.7" "2 1.0 0.5"}; do
  set -- $v
  python3 bench.py --workload cnv --no-cpu --steps 10 --warmup 3 --chunks $1 --first-chunk $2 --last-chunk $3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunks $1 first $2 last $3: ms', d['ms_per_step'], 'value', d['value'], 'mismatch', d['link']['result_mismatches_vs_resident'])"
done
