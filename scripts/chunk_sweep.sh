#!/bin/bash
# development aid (GPU box): the staged step over chunk plans
run() {
  python3 bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  $*: ms', d['ms_per_step'], 'value', d['value'], 'mismatches', d['link']['result_mismatches_vs_resident'])"
}
for c in 6 10 12 16; do run --chunks $c; done
run --chunks 10 --last-chunk 0.3
run --chunks 12 --last-chunk 0.3
for c in "3 0.5" "4 0.5" "4 1.0" "5 0.5"; do set -- $c; run --workload cnv --chunks $1 --first-chunk $2; done
for c in 2 4 5; do run --dnms 12500 --chunks $c; done
