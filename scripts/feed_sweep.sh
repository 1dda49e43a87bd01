#!/bin/bash
# development aid (GPU box): files -> results over the chunk size of the feed pass
for fc in ${FCS:-1500 2500 3400 5000}; do
  python3 bench.py --no-cpu --no-config5 --steps 3 --warmup 2 --feed-chunk $fc 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); f=d['feed']; print('feed-chunk $fc: value_e2e', d['value_e2e'], 'seconds', f['seconds'], 'chunks', f['chunks'], 'stage busy', f['bam_stage_seconds_busy'], 'inflate busy', f['device_inflate']['seconds_busy'], 'mismatches', f['result_mismatches_vs_resident'])"
done
