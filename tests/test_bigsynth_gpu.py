"""Benchmark-scale generator and pipeline on the GPU:
 * the HIP build of synth/uzsynth.h (staged format, written in place in HBM) equals the gcc build (ASCII form)
   packed by the host library;
 * the whole path on HBM-resident (adopted) columns equals the CPU oracle on the host copy -- DNMs placed
   uniformly, so neighbouring DNMs share alignment records;
 * the STAGED path (fetch-reach selection on the host, asynchronous packed upload, chunks of DNMs) gives the
   same per-DNM results as the resident table;
 * size-independent properties at a larger size: calls agree with the simulated truth, a second pass is
   idempotent, and phasing a sub-batch gives the same per-DNM results."""
import numpy as np
import pytest

from synth import bigsynth
from synth.sites_np import make_clusters, make_sites, place_dnms_full
from unfazed_amd import abi, io_native
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
    dn = place_dnms_full(sc, 3000, seed=32, indel_frac=0.2)  # dense: about a third of the DNMs share a cluster
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=33)
    wl = bigsynth.WorkloadOnGpu(cfg, sc, dn, cl, device=0)
    yield sc, dn, cl, cfg, wl
    wl.free()


def test_gpu_generator_equals_cpu_generator(workload):
    sc, dn, cl, cfg, wl = workload
    assert (cl.nd > 1).sum() > 50
    full = wl.download()
    c1 = 60
    rh, _ = bigsynth.reads_cpu(cfg, sc, dn, cl, 0, c1, threads=4)
    want = io_native.pack_reads(rh, cfg.min_base_qual, lists=False, with_end=True)  # the generator writes the quality plane itself
    n = int(want.view.n_segs)
    for name, _ in abi.PACKED_RECORD_COLS:
        assert np.array_equal(want.arrays[name][:n], full.arrays[name][:n]), name
    assert np.array_equal(want.arrays["cigar"][: want.view.n_cigar_total], full.arrays["cigar"][: want.view.n_cigar_total])
    assert want.view.n_exc == 0 and full.view.n_exc == 0  # the generator writes A/C/G/T only
    for name, unit in (("seq2", 8), ("qlow", 4)):
        k = int(want.view.n_row_units) * unit
        assert np.array_equal(want.arrays[name][:k], full.arrays[name][:k]), name


def _oracle(sc, dn, cl, cfg, P, cutoff, d_hi):
    from oracle import oracle as orc
    c_hi = cl.of_dnm(d_hi - 1) + 1
    m = int(cl.d0[c_hi - 1] + cl.nd[c_hi - 1])  # whole clusters
    rh, _ = bigsynth.reads_cpu(cfg, sc, dn, cl, 0, c_hi, threads=4)
    sh, fh = _host_views(sc)
    dvm = abi.dnms_view(dn.contig[:m], dn.contig[:m], dn.start[:m], dn.end[:m], np.zeros(m, np.uint8), dn.refs[:m],
                        dn.alts[:m], cutoff)
    found = orc.find(P, sh, fh, dvm, abi.FIND_SECOND_WINDOW)
    return m, orc.phase(P, sh, rh, dvm, found, keep_lists=True)


def test_resident_pipeline_matches_oracle(workload, engine):
    sc, dn, cl, cfg, wl = workload
    P = abi.make_params()
    engine.set_params(P)
    sid = engine.adopt_sites(wl.sites_view())
    fid = engine.adopt_family(sid, wl.family_view())
    rid = engine.adopt_reads(wl.reads_view())
    n = dn.n
    cutoff = concordant_cutoff(wl.tlen_head(), P.readlen, 3)
    dv = abi.dnms_view(dn.contig, dn.contig, dn.start, dn.end, np.zeros(n, np.uint8), dn.refs, dn.alts, cutoff)
    got = engine.phase_raw(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
    vo, vv = engine.votes(n)
    # oracle on the first clusters, reads regenerated on the CPU (query-name ids = pair numbers: the same in both)
    m, want = _oracle(sc, dn, cl, cfg, P, cutoff, 400)
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

    # the staged path: per chunk of DNMs, the records their fetches can reach (host selection), uploaded
    # asynchronously in the packed form; chunk boundaries fall INSIDE clusters, so records are duplicated across chunks
    from unfazed_amd.engine import PinnedPool
    from unfazed_amd.staging import fetch_points
    full = wl.download()
    src = io_native.ReadsSource(full)
    co, ci, cf, ho, hi = engine.find(fid, dv, P, abi.FIND_SECOND_WINDOW)
    pool = PinnedPool()
    bounds = [0, 701, 1399, 2222, n]
    rids, staged_bytes, maps = [], 0, []
    for a, b in zip(bounds[:-1], bounds[1:]):
        alen = np.array([max(len(r), len(x)) for r, x in zip(dn.refs[a:b], dn.alts[a:b])], np.int64)
        fc, flo, fhi, fex = fetch_points(dn.contig[a:b], dn.start[a:b], np.zeros(b - a, np.uint8), sc.pos, ho[a: b + 1], hi, P, allele_len=alen)
        # chunks alternate between whole rows and the fetched 32-base units only (unit masks); every other chunk is carved from
        # one pinned block, so its columns cross the link as ONE copy into a mirror block instead of one copy per column
        if (a // 700) % 2 == 1:
            pool.new_slab(256 << 20)
        part = src.select(fc, flo, fhi, alloc=pool.alloc, extra=fex if (a // 700) % 2 == 0 else None, tuples=(a // 700) % 3 != 2,
                          start8=(a // 700) % 2 == 1,  # (start differences in eight / sixteen bits)
                          pair8=(a // 700) == 1)       # (tlen / mate / name id in the pair form's one byte, or as eight- / sixteen-bit differences)
        pool.end_slab()
        assert ("pair_d8" in part.arrays) == ((a // 700) == 1) and ("mate_d8" in part.arrays) == ((a // 700) == 3)
        maps.append(part.qname_map)
        staged_bytes += sum(x.nbytes for x in part.arrays.values())
        part.arrays.update(abi.small_columns(part)); part.arrays.update(abi.wide_columns(part))  # (plain views for the checks below; not staged)
        if "umask" in part.arrays:
            um = part.arrays["umask"][: part.view.n_segs]
            with_bases = (part.arrays["aux"][: part.view.n_segs] & abi.AUX_NO_SEQ) == 0
            assert 0.5 < (um[with_bases] != abi.UMASK_ALL).mean() and part.view.n_seq_units < 0.5 * abi.row_units(part.arrays["l_seq"][: part.view.n_segs])[with_bases].sum()
        assert 0.3 < (part.arrays["aux"][: part.view.n_segs] & abi.AUX_NO_SEQ).astype(bool).mean() < 0.6  # mates travel without bases
        rids.append(engine.upload_reads_packed(part))
    assert staged_bytes < 0.3 * sum(x.nbytes for x in full.arrays.values())
    for (a, b), r, qmap in zip(zip(bounds[:-1], bounds[1:]), rids, maps):
        dvc = abi.dnms_view(dn.contig[a:b], dn.contig[a:b], dn.start[a:b], dn.end[a:b], np.zeros(b - a, np.uint8),
                            dn.refs[a:b], dn.alts[a:b], cutoff)
        res = engine.phase_raw(fid, r, dvc, P, abi.FIND_SECOND_WINDOW)
        for k in ("status", "counts", "origin", "evidence"):
            assert np.array_equal(res[k], got[k][a:b]), (k, a, b)
        so, sv_ = engine.votes(b - a)
        for d in range(0, b - a, 7):
            for j in range(4):
                mine = sv_[so[4 * d + j]: so[4 * d + j + 1]]
                if j < 2 and qmap is not None:  # read lists hold name ids: the pair form numbered the chunk's names itself
                    mine = qmap[mine]
                assert np.array_equal(mine, vv[vo[4 * (a + d) + j]: vo[4 * (a + d) + j + 1]]), (a + d, j)
        engine.free_reads(r)
    pool.free_all()
    engine.free_reads(rid)
    engine.free_sites(sid)


def test_read_stage_in_two_halves(workload, engine):
    """uz_phase_begin / uz_phase_end: the batch queued, another batch's window emit run in between, the results handed out afterwards
    -- the same as uz_phase's; a batch that outgrows the sizes it was queued on (those of the batch before it) is run again inside
    uz_phase_end, with the same results."""
    sc, dn, cl, cfg, wl = workload
    P = abi.make_params()
    engine.set_params(P)
    sid = engine.adopt_sites(wl.sites_view())
    fid = engine.adopt_family(sid, wl.family_view())
    rid = engine.adopt_reads(wl.reads_view())
    n = dn.n
    cutoff = concordant_cutoff(wl.tlen_head(), P.readlen, 3)

    def view(sel):
        return abi.dnms_view(dn.contig[sel], dn.contig[sel], dn.start[sel], dn.end[sel], np.zeros(len(sel), np.uint8),
                             [dn.refs[i] for i in sel], [dn.alts[i] for i in sel], cutoff)
    everything = np.arange(n)
    dv = view(everything)
    got = engine.phase_raw(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
    a, b = everything[::2], everything[1::2]
    dva, dvb = view(a), view(b)
    want_a = engine.phase_raw(fid, rid, dva, P, abi.FIND_SECOND_WINDOW)
    va = engine.votes(len(a))
    engine.phase_begin(fid, rid, dva, P, abi.FIND_SECOND_WINDOW)
    co, ci, cf, ho, hi = engine.find(fid, dvb, P, abi.FIND_SECOND_WINDOW)  # another batch's window lists, queued behind the read stage
    assert co[-1] > 0
    res = engine.phase_end(fid, rid, dva, P, abi.FIND_SECOND_WINDOW)
    vb = engine.votes(len(a))
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(res[k], want_a[k]) and np.array_equal(res[k], got[k][a]), k
    assert np.array_equal(va[0], vb[0]) and np.array_equal(va[1], vb[1])
    with pytest.raises(Exception):
        engine.phase_end(fid, rid, dva, P, abi.FIND_SECOND_WINDOW)  # no batch is open
    # a batch of DNMs without candidate sites leaves tiny sizes behind: the whole batch queued on those cannot fit
    none = np.nonzero(got["status"] == abi.ST_NO_CAND)[0]
    assert none.size >= 3
    engine.phase_raw(fid, rid, view(none[:3]), P, abi.FIND_SECOND_WINDOW)
    engine.phase_begin(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
    engine.find(fid, dvb, P, abi.FIND_SECOND_WINDOW)
    again = engine.phase_end(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(again[k], got[k]), k
    engine.free_reads(rid)
    engine.free_sites(sid)
