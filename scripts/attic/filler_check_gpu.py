"""development aid: the device's walk + joins on a BAM with filler between the pile-ups, against the one-pass stage"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from synth import bigsynth
from synth.sites_np import make_clusters, make_sites, place_dnms_full
from unfazed_amd import io_native
from unfazed_amd.engine import HipEngine
eng = HipEngine(0)
d = tempfile.mkdtemp()
lens = [3_000_000, 2_000_000, 1_000_000]
sc = make_sites(12000, seed=77, contig_lens=lens)
dn = place_dnms_full(sc, int(sys.argv[1]) if len(sys.argv) > 1 else 120, seed=78)
cl = make_clusters(dn)
cfg = bigsynth.make_cfg(seed=79); cfg.n_clusters = cl.n
for filler in (0.0, 30.0):
    bam = os.path.join(d, "f%d.bam" % int(filler))
    st = bigsynth.write_bam(bam, cfg, sc, dn, cl, contig_len=lens, level=1, threads=3, filler=filler, filler_reach=65536)
    rng = np.random.default_rng(9)
    c, lo, hi = [], [], []
    for i in range(dn.n):
        c.append(dn.contig[i]); lo.append(dn.start[i] - 1); hi.append(dn.start[i] + 1)
        for p in np.sort(rng.integers(dn.start[i] - 5000, dn.start[i] + 5000, 12)):
            c.append(dn.contig[i]); lo.append(int(p)); hi.append(int(p) + 1)
    fc, flo, fhi = np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32)
    src = io_native.BamSource(bam, threads=3)
    ref = src.select(fc, flo, fhi, 20)
    n = int(ref.view.n_segs)
    voff, qn, mt, bs = io_native.stage_kept_debug(src.lib, ref._stage.ptr, n)
    twin = src.select_kept(fc, flo, fhi, 20, small_tasks=True)
    host = src.select_kept(fc, flo, fhi, 20, walk=eng.bam_walk, release=eng.bam_walk_release, merge=True)
    print("filler", filler, "records", st["records"], "kept", n, "| device walk + host joins:", host.n, "flags", int(np.count_nonzero(host.d_flags)),
          "desc", host.desc.size, "twin desc", twin.desc.size)
    # descriptors: device vs twin (restricted to what the device's filter keeps)
    tv = set(twin.desc["voff"].tolist()); dv = set(host.desc["voff"].tolist())
    print("   device descriptors not in the twin's:", len(dv - tv), " kept voffs missing from the device's descriptors:", len(set(voff.tolist()) - dv))
    dev = src.select_kept(fc, flo, fhi, 20, join=eng, release=eng.bam_walk_release)
    got = eng.join_fetch(dev.token, dev.n, len(src.contigs))
    print("   device joins: n", dev.n, "voff equal", dev.n == n and np.array_equal(got["voff"], voff), "qname", dev.n == n and np.array_equal(got["qname"], qn),
          "mate", dev.n == n and np.array_equal(got["mate"], mt))
    eng.bam_walk_release(dev.token); dev.token = None
    del host
