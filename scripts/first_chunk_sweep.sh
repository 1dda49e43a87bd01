run() {
  python3 bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 12 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  $*: ms', d['ms_per_step'], 'mismatches', d['link']['result_mismatches_vs_resident'])"
}
for c in "3 1.0" "3 0.5" "3 0.35" "4 0.5" "4 0.35" "2 0.5"; do set -- $c; run --dnms 12500 --chunks $1 --first-chunk $2; done
for c in "3 1.0" "3 0.5" "4 0.5" "4 0.35" "5 0.5"; do set -- $c; run --workload cnv --chunks $1 --first-chunk $2; done
for c in "6 1.0" "6 0.5" "7 0.5"; do set -- $c; run --chunks $1 --first-chunk $2; done
