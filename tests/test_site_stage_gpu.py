"""GPU parity of the site stage (K1 site scan, K2 window emit) against the CPU
oracle on the same seeded columns, through the C ABI.  Bit-exact (integer/byte)."""
import numpy as np
import pytest

from synth.sites_np import make_sites, place_dnms
from unfazed_amd import abi

pytestmark = pytest.mark.gpu


def _views(sc):
    sv = abi.SitesView()
    arrs = dict(contig_off=sc.contig_off, pos=sc.pos, sflags=sc.sflags, ref_base=sc.ref_base, alt_base=sc.alt_base)
    sv.n_sites = sc.n
    sv.n_contigs = len(sc.contig_off) - 1
    for k, a in arrs.items():
        setattr(sv, k, a.ctypes.data)
    return abi.Held(sv, arrs), abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)


class _Sites:
    """minimal object the engine's upload_sites accepts"""
    def __init__(self, sc):
        self.contig_off, self.pos, self.sflags, self.ref_base, self.alt_base = sc.contig_off, sc.pos, sc.sflags, sc.ref_base, sc.alt_base
        self.contigs = sc.contig_names
        self.n_sites = sc.n


PARAM_SETS = [
    dict(),
    dict(min_gt_qual=30, min_depth=0, ab_het=(0.3, 0.7)),
    dict(min_gt_qual=0, min_depth=25, ab_homref=(0.0, 0.05), ab_homalt=(0.95, 1.0), ab_het=(0.45, 0.55)),
    dict(min_gt_qual=-5, min_depth=-5),
]


@pytest.mark.parametrize("n_sites", [0, 1, 7, 8, 9, 4097, 300_001])
def test_site_scan_matches_oracle(engine, n_sites):
    from oracle import oracle as orc
    sc = make_sites(max(n_sites, 1), seed=7 + n_sites, contig_lens=[5e6, 3e6, 1e6], weird_frac=0.08, complex_frac=0.05)
    if n_sites == 0:
        for name in ("pos", "sflags", "ref_base", "alt_base", "gt"):
            setattr(sc, name, getattr(sc, name)[:0])
        sc.rd, sc.ad, sc.gq = sc.rd[:, :0].copy(), sc.ad[:, :0].copy(), sc.gq[:, :0].copy()
        sc.contig_off = np.zeros_like(sc.contig_off)
    sh, fh = _views(sc)
    sid = engine.upload_sites(_Sites(sc))
    fid = engine.add_family(sid, sc.gt, sc.rd, sc.ad, sc.gq)
    for kw in PARAM_SETS:
        P = abi.make_params(**kw)
        want = orc.classify(P, sh, fh)
        got = engine.classify(fid, P, sc.n)
        assert np.array_equal(want, got), (kw, np.nonzero(want != got)[0][:10])
    engine.free_sites(sid)


@pytest.mark.parametrize("mode", [0, abi.FIND_SECOND_WINDOW, abi.FIND_WHOLE_REGION])
def test_window_emit_matches_oracle(engine, mode):
    from oracle import oracle as orc
    sc = make_sites(400_000, seed=11, contig_lens=[4e7, 2e7, 1e7], weird_frac=0.03)
    idx, contig, start, end = place_dnms(sc, 3000, seed=5)
    rng = np.random.default_rng(3)
    n = len(start)
    vt = np.zeros(n, dtype=np.uint8)
    mult = np.ones(n, dtype=np.uint8)
    if mode & abi.FIND_WHOLE_REGION:
        vt = rng.integers(1, 4, n).astype(np.uint8)
        end = (start + rng.integers(1000, 200000, n)).astype(np.int32)
        sd = 0
    else:
        # long events exercise the second window (and overlapping windows), mult the find_many duplicates
        long = rng.random(n) < 0.2
        end = np.where(long, start + rng.integers(4000, 20000, n), end).astype(np.int32)
        mult = np.where(rng.random(n) < 0.1, 2, 1).astype(np.uint8)
        sd = 5000
    contig = contig.copy()
    contig[::97] = -1  # contig missing from the sites file
    start[5] = 10  # window clipped at the contig start
    end[5] = 11
    sh, fh = _views(sc)
    sid = engine.upload_sites(_Sites(sc))
    fid = engine.add_family(sid, sc.gt, sc.rd, sc.ad, sc.gq)
    P = abi.make_params(search_dist=sd)
    dv = abi.dnms_view(contig, [-1] * n, start, end, vt, [b""] * n, [b""] * n, 0.0, mult=mult)
    want = orc.find(P, sh, fh, dv, mode)
    got = engine.find(fid, dv, P, mode)
    for a, b, name in zip(want, got, ("cand_off", "cand_idx", "cand_flags", "het_off", "het_idx")):
        assert np.array_equal(a, b), name
    assert want[0][-1] > 100 and want[3][-1] > 100
    # the fill pass first runs on the room the lists already have (no host wait for the totals in front of it): the same batch again
    # -- now the room is there --, with a room that holds only some of the DNMs' slices (test hook: the second fill must repair it),
    # and a smaller batch behind a larger one (stale entries beyond its totals must not leak)
    import os
    engine.drop_derived()
    again = engine.find(fid, dv, P, mode)
    os.environ["UZ_TEST_FIND_CAP"] = str(int(want[3][-1]) // 3)
    try:
        engine.drop_derived()
        short = engine.find(fid, dv, P, mode)
    finally:
        del os.environ["UZ_TEST_FIND_CAP"]
    m = n // 2
    dv2 = abi.dnms_view(contig[:m], [-1] * m, start[:m], end[:m], vt[:m], [b""] * m, [b""] * m, 0.0, mult=mult[:m])
    want2 = orc.find(P, sh, fh, dv2, mode)
    half = engine.find(fid, dv2, P, mode)
    for name, a, b, c2, w2, h2 in zip(("cand_off", "cand_idx", "cand_flags", "het_off", "het_idx"), want, again, short, want2, half):
        assert np.array_equal(a, b), name + " (second run)"
        assert np.array_equal(a, c2), name + " (room too small)"
        assert np.array_equal(w2, h2), name + " (smaller batch)"
    engine.free_sites(sid)


@pytest.mark.parametrize("n_sites,n_fam", [(4097, 1), (50_003, 5), (300_001, 3)])
def test_cohort_scan_matches_per_family_scans(engine, n_sites, n_fam):
    """uz_site_scan_many (one launch over the families of a sites table) gives every
    family the class bytes of its own uz_site_scan, and those match the oracle."""
    from oracle import oracle as orc
    base = make_sites(n_sites, seed=91, contig_lens=[4e6, 2e6], weird_frac=0.05, complex_frac=0.04)
    sid = engine.upload_sites(_Sites(base))
    fams, cols = [], []
    for k in range(n_fam):  # other trios of the same table: the table's records, other genotype columns
        sc = base if k == 0 else make_sites(n_sites, seed=91 + 7 * k, contig_lens=[4e6, 2e6], weird_frac=0.05, complex_frac=0.04)
        cols.append((sc.gt, sc.rd, sc.ad, sc.gq))
        fams.append(engine.add_family(sid, sc.gt, sc.rd, sc.ad, sc.gq))
    P = abi.make_params(min_gt_qual=15)
    engine.set_params(P)
    engine.site_scan_many(fams)
    got_many = [engine.classify(f, P, n_sites).copy() for f in fams]  # classes are fresh: no rescan
    for f, (gt, rd, ad, gq), gm in zip(fams, cols, got_many):
        engine.site_scan(f)
        assert np.array_equal(gm, engine.classify(f, P, n_sites))
        sh, _ = _views(base)
        want = orc.classify(P, sh, abi.family_view(gt, rd, ad, gq))
        assert np.array_equal(want, gm)
    with pytest.raises(Exception):
        engine.site_scan_many([fams[0], fams[0]])
    engine.free_sites(sid)


@pytest.mark.parametrize("slab", [False, True], ids=["column_copies", "one_slab_copy"])
def test_asynchronous_site_stage_equals_the_synchronous_one(engine, slab):
    """uz_sites_family_upload_async: sites + genotype columns queued on the copy stream, the first use waits for them and folds
    the complex flag -- the same classes and the same windows as the synchronous uploads.  slab: the host columns sit back to
    back in one pinned block and cross the link as ONE copy into a mirror block (abi.hip: SlabPlan)."""
    from unfazed_amd.engine import PinnedPool
    sc = make_sites(200_000, seed=21, contig_lens=[4e7, 2e7], weird_frac=0.03)
    idx, contig, start, end = place_dnms(sc, 1500, seed=6)
    n = len(start)
    P = abi.make_params()
    dv = abi.dnms_view(contig, [-1] * n, start, end, np.zeros(n, np.uint8), [b""] * n, [b""] * n, 0.0)
    sid = engine.upload_sites(_Sites(sc))
    fid = engine.add_family(sid, sc.gt, sc.rd, sc.ad, sc.gq)
    want = engine.find(fid, dv, P, 0)
    want_cls = engine.classify(fid, P, sc.n)
    engine.free_sites(sid)
    pool = PinnedPool()
    if slab:
        pool.new_slab(sc.n * 40 + (1 << 20))

    def pin(a):
        a = np.ascontiguousarray(a)
        out = pool.alloc(max(64, a.nbytes))[: a.nbytes].view(a.dtype).reshape(a.shape)
        out[...] = a
        return out
    cols = {k: pin(getattr(sc, k)) for k in ("pos", "sflags", "ref_base", "alt_base")}
    cols["contig_off"] = pin(np.asarray(sc.contig_off, np.int64))
    gt = pin(sc.gt)
    rd, ad, gq = ([pin(x[m]) for m in range(3)] for x in (sc.rd, sc.ad, sc.gq))
    sv = abi.SitesView()
    sv.n_sites, sv.n_contigs = sc.n, len(sc.contig_off) - 1
    for k, a in cols.items():
        setattr(sv, k, a.ctypes.data)
    for _ in range(2):  # twice: blocks (and the mirror) go back to the pool and are reused
        sid2, fid2 = engine.upload_sites_family_async(abi.Held(sv, cols), gt, rd, ad, gq)
        got = engine.find(fid2, dv, P, 0)
        for a, b, name in zip(want, got, ("cand_off", "cand_idx", "cand_flags", "het_off", "het_idx")):
            assert np.array_equal(a, b), name
        assert np.array_equal(want_cls, engine.classify(fid2, P, sc.n))
        engine.free_sites(sid2)
    assert np.array_equal(gt, sc.gt)  # the host copy is never written (the complex flag is folded into the device's copy)
    pool.free_all()
