run() {
  python3 bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  $*: ms', d['ms_per_step'], 'value', d['value'], 'mismatches', d['link']['result_mismatches_vs_resident'])"
}
for l in 0.3 0.5 0.7 1.0; do run --dnms 12500 --last-chunk $l; done
for l in 0.3 0.7; do run --last-chunk $l; done
