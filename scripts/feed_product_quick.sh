cd $GRAFT_REPO_ROOT
for r in 1 2; do
python3 bench.py --no-cpu --no-config5 --no-staged --steps 1 --warmup 0 --feed-filler-dnms 0 2>/dev/null | grep "^{" | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); f=j['feed']; print(f['value_e2e'], f['seconds_of_every_pass'], 'cpu', f['host_cpu_seconds_per_pass'], 'sites busy', f['sites_decode']['seconds_busy'], 'mism', f['result_mismatches_vs_resident'], 'product', f['product']['value_e2e'], f['product']['seconds_of_every_call'], f['product']['record_mismatches_vs_resident'])"
done
