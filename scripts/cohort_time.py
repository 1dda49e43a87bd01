"""K1 over a cohort: F trios of one sites table in one launch vs F launches (development aid, GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from synth.sites_np import make_sites
from unfazed_amd import abi
from unfazed_amd.engine import HipEngine, K_SITE_SCAN
from test_site_stage_gpu import _Sites

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = make_sites(S, seed=202)
e = HipEngine(0)
sid = e.upload_sites(_Sites(sc))
rng = np.random.default_rng(1)
fams = []
for k in range(F):  # same columns rolled: content does not matter for timing, distinct buffers do
    fams.append(e.add_family(sid, np.roll(sc.gt, k), np.roll(sc.rd, k, axis=1), np.roll(sc.ad, k, axis=1), np.roll(sc.gq, k, axis=1)))
e.set_params(abi.make_params())
e.prof_enable(True)
for rep in range(3):
    e.prof_reset()
    for f in fams:
        e.site_scan(f)
    e.sync()
    ms, n = e.prof_get(K_SITE_SCAN)
    print("F=%d separate launches: %.1f us per family -> %.2f TB/s" % (F, ms / F * 1e3, 20.0 * S * F / (ms * 1e-3) / 1e12), flush=True)
    e.prof_reset()
    e.site_scan_many(fams)
    e.sync()
    ms, n = e.prof_get(K_SITE_SCAN)
    print("F=%d one launch:        %.1f us per family -> %.2f TB/s" % (F, ms / F * 1e3, 20.0 * S * F / (ms * 1e-3) / 1e12), flush=True)
