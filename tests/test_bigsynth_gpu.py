"""Benchmark-scale generator and pipeline on the GPU:
 * the HIP build of synth/uzsynth.h writes the same columns as the gcc build;
 * the whole path on HBM-resident (adopted) columns equals the CPU oracle on the host copy;
 * size-independent properties at a larger size: calls agree with the simulated truth,
   a second pass is idempotent, and phasing a sub-batch gives the same per-DNM results."""
import numpy as np
import pytest

from synth import bigsynth
from synth.sites_np import make_sites, place_dnms_full
from unfazed_amd import abi
from unfazed_amd.hostpath import concordant_cutoff

pytestmark = pytest.mark.gpu


def _host_views(sc):
    sv = abi.SitesView()
    keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=sc.pos, sflags=sc.sflags,
                ref_base=sc.ref_base, alt_base=sc.alt_base)
    sv.n_sites, sv.n_contigs = sc.n, len(sc.contig_off) - 1
    for k, a in keep.items():
        setattr(sv, k, a.ctypes.data)
    return abi.Held(sv, keep), abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)


@pytest.fixture(scope="module")
def workload():
    sc = make_sites(400_000, seed=31, contig_lens=[6e7, 4e7, 2e7])
    dn = place_dnms_full(sc, 1500, seed=32, indel_frac=0.2)
    cfg = bigsynth.make_cfg(seed=33, n_pairs=1200, half_width=6000, n_dnms=dn.n)
    wl = bigsynth.WorkloadOnGpu(cfg, sc, dn, device=0)
    yield sc, dn, cfg, wl
    wl.free()


def test_gpu_generator_equals_cpu_generator(workload):
    sc, dn, cfg, wl = workload
    m = 40
    _, cpu = bigsynth.reads_cpu(cfg, sc, dn, 0, m)
    gpu = wl.download_block(0, m)
    for name, _, _ in bigsynth.OUT_COLS:
        assert np.array_equal(cpu[name], gpu[name]), name


def test_resident_pipeline_matches_oracle(workload, engine):
    from oracle import oracle as orc
    sc, dn, cfg, wl = workload
    P = abi.make_params()
    engine.set_params(P)
    sid = engine.adopt_sites(wl.sites_view())
    fid = engine.adopt_family(sid, wl.family_view())
    rid = engine.adopt_reads(wl.reads_view())
    n = dn.n
    head = wl.dev.get(wl.out_ptrs["tlen"], (wl.n_segs,), np.int32)
    cutoff = concordant_cutoff(head, P.readlen, 3)
    dv = abi.dnms_view(dn.contig, dn.contig, dn.start, dn.end, np.zeros(n, np.uint8), dn.refs, dn.alts, cutoff)
    got = engine.phase_raw(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
    vo, vv = engine.votes(n)
    # oracle on the first m DNMs, reads regenerated on the CPU
    m = 300
    rh, _ = bigsynth.reads_cpu(cfg, sc, dn, 0, m)
    sh, fh = _host_views(sc)
    dvm = abi.dnms_view(dn.contig[:m], dn.contig[:m], dn.start[:m], dn.end[:m], np.zeros(m, np.uint8), dn.refs[:m],
                        dn.alts[:m], cutoff)
    found = orc.find(P, sh, fh, dvm, abi.FIND_SECOND_WINDOW)
    want = orc.phase(P, sh, rh, dvm, found, keep_lists=True)
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(want[k], got[k][:m]), k
    wo, wv = want["vote_off"], want["vote_val"]
    for d in range(m):
        for j in range(4):
            assert np.array_equal(wv[wo[4 * d + j]: wo[4 * d + j + 1]], vv[vo[4 * d + j]: vo[4 * d + j + 1]]), (d, j)
    assert (want["status"] == abi.ST_OK).sum() > 20
    # properties at the full size -------------------------------------------------------
    called = (got["status"] == abi.ST_OK) & ((got["origin"] == abi.OR_DAD) | (got["origin"] == abi.OR_MOM))
    truth = np.where(dn.origin == 0, abi.OR_DAD, abi.OR_MOM)
    assert called.sum() > 100
    assert (got["origin"][called] == truth[called]).mean() > 0.97
    # idempotence of a second pass from the staged inputs
    engine.drop_derived()
    again = engine.phase_raw(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(again[k], got[k]), k
    # a sub-batch (every third DNM) gives the same per-DNM results: DNMs are independent
    sel = np.arange(0, n, 3)
    dvs = abi.dnms_view(dn.contig[sel], dn.contig[sel], dn.start[sel], dn.end[sel], np.zeros(len(sel), np.uint8),
                        [dn.refs[i] for i in sel], [dn.alts[i] for i in sel], cutoff)
    sub = engine.phase_raw(fid, rid, dvs, P, abi.FIND_SECOND_WINDOW)
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(sub[k], got[k][sel]), k
    engine.free_reads(rid)
    engine.free_sites(sid)
