#!/bin/bash
# The product's drop-in call (bench.py feed.product) over BAM stages in flight x DNMs per chunk (GPU box, through gpurun).
# usage: scripts/product_sweep.sh   -> one line per setting
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for ahead in ${AHEADS:-2 3 4}; do for chunk in ${CHUNKS:-3400 5000}; do
  UZ_HOST_AHEAD=$ahead UZ_HOST_CHUNK_DNMS=$chunk python bench.py --no-cpu --no-config5 --no-staged --steps 1 --warmup 0 2>/dev/null | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());p=d['feed']['product'];print('ahead $ahead chunk $chunk: product %.3f s = %.1f k DNMs/s, mismatches %d; feed %.3f s' % (p['seconds'], p['value_e2e']/1e3, p['record_mismatches_vs_resident'], d['feed']['seconds']))"
done; done
