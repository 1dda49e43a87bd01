"""Quick K1 timing on the GPU box (development aid)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from synth.sites_np import make_sites, place_dnms
from unfazed_amd import abi
from unfazed_amd.engine import HipEngine, K_SITE_SCAN, K_WINDOW_COUNT, K_WINDOW_FILL
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_site_stage_gpu import _Sites

S = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
t = time.time()
sc = make_sites(S, seed=202)
idx, contig, start, end = place_dnms(sc, N, seed=201)
print("gen %.1fs" % (time.time() - t), flush=True)
e = HipEngine(0)
sid = e.upload_sites(_Sites(sc))
fid = e.add_family(sid, sc.gt, sc.rd, sc.ad, sc.gq)
P = abi.make_params()
e.set_params(P)
e.prof_enable(True)
for rep in range(3):
    e.prof_reset()
    for i in range(20):
        e.site_scan(fid)
    e.sync()
    ms, n = e.prof_get(K_SITE_SCAN)
    us = ms / n * 1e3
    print("K1: %.1f us/launch  -> %.2f TB/s at 20 B/site (%d sites)" % (us, 20.0 * S / (us * 1e-6) / 1e12, S), flush=True)
dv = abi.dnms_view(contig, [-1] * N, start, end, np.zeros(N, np.uint8), [b""] * N, [b""] * N, 0.0)
e.prof_reset()
t = time.time()
for i in range(5):
    co, ci, cf, ho, hi = e.find(fid, dv, P, abi.FIND_SECOND_WINDOW, fetch=False)
e.sync()
dt = (time.time() - t) / 5
print("find: %.2f ms wall/batch; count %.3f ms fill %.3f ms; cand %d het %d" % (dt * 1e3, e.prof_get(K_WINDOW_COUNT)[0] / 5, e.prof_get(K_WINDOW_FILL)[0] / 5, co[-1], ho[-1]))
