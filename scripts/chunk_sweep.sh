#!/bin/bash
# development aid (GPU box): the staged step over chunk plans.  scripts/chunk_sweep.sh [snv|shard|cnv ...]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd "$ROOT"
run() {
  python3 bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  $*: ms', d['ms_per_step'], 'value', d['value'], 'mismatches', d['link']['result_mismatches_vs_resident'])"
}
for what in "${@:-snv}"; do
  case $what in
    snv) for c in 4 5 6 7 8; do run --chunks $c; done; run --chunks 5 --first-chunk 1.0; run --chunks 6 --first-chunk 1.0;;
    shard) for c in 2 3 4; do run --dnms 12500 --chunks $c; done; run --dnms 12500 --chunks 2 --first-chunk 1.0 --last-chunk 1.0;;
    cnv) for c in "3 0.5" "4 0.5" "4 1.0" "5 0.5"; do set -- $c; run --workload cnv --chunks $1 --first-chunk $2; done;;
  esac
done
