"""Host overhead of the drop-in call phase_snvs(...) at scale, device excluded: the same host code (unfazed_amd.hostpath) driven by a
stand-in backend that answers from numpy in microseconds, so that what is timed is the Python around the C ABI -- the filters in the
reference's order, the DNM batch, the records dict.   python scripts/host_overhead.py [n_dnms] [--profile]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from synth.sites_np import make_sites, place_dnms_full  # noqa: E402
from unfazed_amd import abi, session  # noqa: E402
from unfazed_amd.model import ReadsTable, SitesTable  # noqa: E402
from unfazed_amd.snv_phaser import phase_snvs  # noqa: E402


class StandIn:
    """answers like the device would, from numpy: window lists by searchsorted, a third of the DNMs phased"""

    def __init__(self, pos, contig_off):
        self.pos, self.contig_off = pos, contig_off

    def upload_sites(self, sites):
        return 0

    def add_family(self, *a):
        return 0

    def upload_reads(self, *a, **k):
        return 0

    def find(self, fam, dv, params, mode):
        a = dv.arrays
        n = dv.view.n
        c = np.asarray(a["contig"][:n], np.int64)
        st = np.asarray(a["start"][:n], np.int64)
        lo = np.zeros(n, np.int64)
        hi = np.zeros(n, np.int64)
        for cc in np.unique(c):
            if cc < 0:
                continue
            m = c == cc
            seg = self.pos[self.contig_off[cc]: self.contig_off[cc + 1]]
            lo[m] = self.contig_off[cc] + np.searchsorted(seg, st[m] - 5000)
            hi[m] = self.contig_off[cc] + np.searchsorted(seg, st[m] + 5000)
        cnt = hi - lo
        ho = np.zeros(n + 1, np.int64)
        ho[1:] = np.cumsum(cnt)
        idx = (np.repeat(lo, cnt) + (np.arange(int(cnt.sum())) - np.repeat(ho[:-1], cnt))).astype(np.int32)
        keep = (np.arange(idx.size) % 3) == 0  # a third of them candidates
        co = np.zeros(n + 1, np.int64)
        co[1:] = np.cumsum(np.add.reduceat(keep.astype(np.int64), ho[:-1]) if idx.size else 0)
        return co, idx[keep], np.zeros(int(keep.sum()), np.uint8), ho, idx

    def phase(self, fam, rh, dv, params, found, want_lists=True, find_mode=2):
        n = dv.view.n
        rng = np.random.default_rng(1)
        status = np.where(rng.random(n) < 0.33, abi.ST_OK, abi.ST_NO_OVERLAP).astype(np.int32)
        counts = rng.integers(0, 6, (n, 4)).astype(np.int32)
        lists = None
        if want_lists:
            e = np.zeros(0, np.int32)
            lists = [(np.arange(c[0], dtype=np.int32), np.arange(c[1], dtype=np.int32), np.arange(c[2], dtype=np.int32) + 100, np.arange(c[3], dtype=np.int32) + 200)
                     if s == abi.ST_OK else (e, e, e, e) for s, c in zip(status, counts)]
        return dict(status=status, counts=counts, origin=np.zeros(n, np.int32), evidence=np.zeros(n, np.int32), lists=lists)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
    sc = make_sites(max(200000, n * 200), seed=202)
    dn = place_dnms_full(sc, n, seed=201)
    t = SitesTable(["kid", "dad", "mom"], sc.contig_names)
    t.contig_off, t.pos, t.end = np.asarray(sc.contig_off, np.int64), sc.pos, sc.pos + 1
    t.sflags, t.ref_base, t.alt_base = sc.sflags, sc.ref_base, sc.alt_base
    t.gt = np.stack([(sc.gt >> (2 * m)) & 3 for m in range(3)]).astype(np.uint8)
    t.ref_depth = np.stack([np.where(sc.rd[m] == 0xFFFF, -1, sc.rd[m].astype(np.int32)).astype(np.int32) for m in range(3)])
    t.alt_depth = np.stack([np.where(sc.ad[m] == 0xFFFF, -1, sc.ad[m].astype(np.int32)).astype(np.int32) for m in range(3)])
    t.gq = np.stack([np.where(sc.gq[m] == 0xFFFF, -1.0, sc.gq[m].astype(np.float64)) for m in range(3)])

    class Strs:
        def __init__(self, col, wrap):
            self.col, self.wrap = col, wrap

        def __len__(self):
            return len(self.col)

        def __getitem__(self, i):
            s = chr(int(self.col[i])) if self.col[i] else "AT"
            return [s] if self.wrap else s
    t.ref_str, t.alt_strs = Strs(sc.ref_base, False), Strs(sc.alt_base, True)
    rt = ReadsTable(sc.contig_names)
    rt.tlen_head = np.full(1000, 450, np.int32)
    rt.qnames = ["r%d" % i for i in range(1000)]
    session.register_sites("S", t)
    session.register_reads("kid.bam", rt)
    session.set_backend(StandIn(sc.pos, np.asarray(sc.contig_off, np.int64)))
    ped = {"kid": {"kid": "kid", "dad": "dad", "mom": "mom", "sex": "2"}}
    dnms = [dict(chrom=sc.contig_names[int(c)], start=int(s), end=int(e), kid="kid", vartype="POINT", bam="kid.bam", cram_ref=None)
            for c, s, e in zip(dn.contig, dn.start, dn.end)]
    args = (["kid"], ped, "S", 2, "38", False, 10 ** 9, True, [0.0, 0.2], [0.8, 1.0], [0.2, 0.8], 20, 10, 5000, 1000000, 3, 1, 151, 5)

    def run():
        d2 = [dict(x) for x in dnms]
        t0 = time.perf_counter()
        recs = phase_snvs(d2, *args)
        return time.perf_counter() - t0, len(recs)
    run()
    if "--profile" in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        run()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    dt, nr = min(run() for _ in range(3))
    print("phase_snvs host path: %d DNMs -> %d records in %.3f s = %.0f DNMs/s (%.1f us per DNM), device answers excluded" % (n, nr, dt, n / dt, dt / n * 1e6))


main()
