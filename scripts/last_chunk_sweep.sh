#!/bin/bash
# development aid (GPU box): the staged step of the 100 k-DNM pass over the size of the last chunk (alternating, to see past the noise)
run() {
  python3 bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  $*: ms', d['ms_per_step'], 'value', d['value'], 'mismatches', d['link']['result_mismatches_vs_resident'])"
}
for rep in 1 2 3; do for l in ${LASTS:-0.5 0.7 1.0}; do run --last-chunk $l; done; done
