cd ${GRAFT_REPO_ROOT:-/root/repo}
run() {
  python3 bench.py --no-cpu --feed-dnms 0 --no-config5 --steps 20 --warmup 5 "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  $*: ms', d['ms_per_step'], 'resident', d['ms_per_step_resident'], 'cpu', d['host_cpu_seconds_per_step'], 'mism', d['link']['result_mismatches_vs_resident'])"
}
run --dnms 12500 --chunks 1
run --dnms 12500 --chunks 2 --first-chunk 1.0 --last-chunk 1.0
run --dnms 12500 --chunks 2 --first-chunk 1.0 --last-chunk 0.6
run --dnms 12500 --chunks 3 --first-chunk 0.5 --last-chunk 0.7
UZ_PIPE_LAG=1 run --dnms 12500 --chunks 3 --first-chunk 1.0 --last-chunk 1.0
run --dnms 12500 --chunks 4 --first-chunk 0.5 --last-chunk 0.5
