"""development aid: wall time of the two calls of a config-5 resident step (read stage / allele balance), iteration by iteration"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from synth import bigsynth
from synth.sites_np import breakpoint_dnms, make_clusters, make_sites, place_cnvs
from unfazed_amd import abi
from unfazed_amd.engine import HipEngine
from unfazed_amd.hostpath import concordant_cutoff
sc = make_sites(20000000, seed=202)
ev = place_cnvs(sc, 10000, seed=501, redraw_seed=502)
dn = breakpoint_dnms(ev); cl = make_clusters(dn); cfg = bigsynth.make_cfg(seed=203)
wl = bigsynth.WorkloadOnGpu(cfg, sc, dn, cl, device=0)
eng = HipEngine(0); P = abi.make_params(); eng.set_params(P)
sid = eng.adopt_sites(wl.sites_view()); fid = eng.adopt_family(sid, wl.family_view()); rid = eng.adopt_reads(wl.reads_view())
cutoff = concordant_cutoff(wl.tlen_head(), P.readlen, 3)
n = ev.n
dv = abi.dnms_view(ev.contig, ev.contig, ev.start, ev.end, ev.vartype, [b""] * n, [b""] * n, cutoff)
for it in range(8):
    eng.drop_derived(); eng.sync()
    t0 = time.perf_counter(); r = eng.phase_raw(fid, rid, dv, P, abi.FIND_SECOND_WINDOW); t1 = time.perf_counter()
    k = eng.phase_cnv(fid, dv, P, rb_counts=r["counts"], want_lists=False); t2 = time.perf_counter()
    print("iter %d: phase_raw %.2f ms, phase_cnv %.2f ms" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
