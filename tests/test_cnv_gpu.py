"""K6 on the device (uz_phase_cnv: whole-region window emit + allele-balance count + summarize_record's decision) against
the oracle's restatement of phase_by_snvs / summarize_record, on the config-5 style workload:
DEL / DUP events of 1 kb - 300 kb with re-drawn interior genotypes, with and without read-backed counts to merge."""
import numpy as np
import pytest

from oracle import oracle as orc
from synth.sites_np import make_sites, place_cnvs
from unfazed_amd import abi

pytestmark = pytest.mark.gpu


def _views(sc):
    sv = abi.SitesView()
    keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=sc.pos, sflags=sc.sflags,
                ref_base=sc.ref_base, alt_base=sc.alt_base)
    sv.n_sites, sv.n_contigs = sc.n, len(sc.contig_off) - 1
    for k, a in keep.items():
        setattr(sv, k, a.ctypes.data)
    return abi.Held(sv, keep), abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)


class _T:  # the attributes engine.upload_sites reads
    def __init__(self, sc):
        self.contig_off, self.pos, self.sflags, self.ref_base, self.alt_base = sc.contig_off, sc.pos, sc.sflags, sc.ref_base, sc.alt_base
        self.n_sites = sc.n
        self.contigs = [str(i) for i in range(len(sc.contig_off) - 1)]


@pytest.mark.parametrize("n_events,params", [(3000, {}), (800, dict(min_depth=4, ab_het=(0.1, 0.9), evidence_min_ratio=2))])
def test_cnv_stage_matches_oracle(engine, n_events, params):
    sc = make_sites(6_000_000, seed=77)
    cv = place_cnvs(sc, n_events, seed=78, redraw_seed=79)
    n = cv.n
    P = abi.make_params(**params)
    sh, fh = _views(sc)
    sid = engine.upload_sites(_T(sc))
    fid = engine.add_family(sid, sc.gt, sc.rd, sc.ad, sc.gq)
    vt = cv.vartype.copy()
    vt[::17] = abi.VT_OTHER_SV  # not DEL / DUP: never phased by allele balance (sv_phaser.py:401)
    vt[5::23] = abi.VT_POINT
    dv = abi.dnms_view(cv.contig, cv.contig, cv.start, cv.end, vt, [b""] * n, [b""] * n, 0.0)
    rng = np.random.default_rng(1)
    rb = rng.choice([0, 0, 0, 1, 2, 10, 11, 30], size=(n, 4)).astype(np.int32)
    for rb_counts in (None, rb):
        want = orc.phase_cnv(P, sh, fh, dv, rb_counts)
        got = engine.phase_cnv(fid, dv, P, rb_counts)
        for k in ("cnv_counts", "origin", "evidence", "etype"):
            assert np.array_equal(want[k], got[k]), k
        for d in range(n):
            for j in range(2):
                assert np.array_equal(want["lists"][d][j], got["lists"][d][j]), (d, j)
    c = want["cnv_counts"]
    assert (c.sum(1) > 0).sum() > n // 3 and c.max() > 40
    # against the simulated truth (allele-balance only): the called parent is the one the event was placed on
    r = engine.phase_cnv(fid, dv, P, None, want_lists=False)
    called = ((r["origin"] == abi.OR_DAD) | (r["origin"] == abi.OR_MOM)) & ((r["etype"] & abi.ET_AMBIG_FLAG) == 0)
    truth = np.where(cv.origin == 0, abi.OR_DAD, abi.OR_MOM)
    assert called.sum() > n // 4
    assert (r["origin"][called] == truth[called]).mean() > 0.97
    engine.free_sites(sid)
