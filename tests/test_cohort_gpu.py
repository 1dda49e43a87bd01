"""Cohort batches (uz_phase_cohort): the DNMs of several kids -- own trio columns, own alignment
records, own insert cutoff each -- in one launch sequence give, DNM by DNM, what one uz_phase per kid gives, vote lists
(query-name ids of the kid's own table) included; and the host path, which routes multi-kid batches through it,
reproduces the reference's two-kid goldens (tests/test_golden_gpu.py runs those too)."""
import numpy as np
import pytest

from helpers import tables
from synth.small import SmallConfig, make_small
from unfazed_amd import abi, io_native
from unfazed_amd.hostpath import concordant_cutoff

pytestmark = pytest.mark.gpu


def test_cohort_batch_equals_per_kid_batches(engine):
    kids = ["kidA", "kidB", "kidC"]
    ds = make_small(SmallConfig(seed=909, n_dnms=21, kids=kids, cluster_prob=0.6, odd_read_prob=0.05))
    sites, reads = tables(ds)
    P = abi.make_params()
    engine.set_params(P)
    sid = engine.upload_sites(sites)
    fams, rhs, cuts, per_kid = {}, {}, {}, {}
    for kid in kids:
        ped = ds.pedigrees[kid]
        fams[kid] = engine.add_family(sid, *sites.family_columns(kid, ped["dad"], ped["mom"]))
        rt = reads["mem://%s.bam" % kid]
        # one kid staged in the packed form, the others in the ASCII form: the merged table takes either
        rhs[kid] = engine.upload_reads(rt, min_base_qual=P.min_gt_qual if kid == "kidB" else None)
        cuts[kid] = concordant_cutoff(rt.tlen, P.readlen, 3) + (7.0 if kid == "kidC" else 0.0)  # cutoffs differ per kid

    def view(dn, kid, cutoff):
        rt = reads["mem://%s.bam" % kid]
        refs, alts = [], []
        for d in dn:
            j = int(sites.query(d["chrom"], d["start"], d["start"] + 1)[-1])
            refs.append(sites.ref_str[j].encode())
            alts.append(sites.alt_strs[j][0].encode())
        return (dict(contig=[sites.contig_index[d["chrom"]] for d in dn], rcontig=[rt.contig_index[d["chrom"]] for d in dn],
                     start=[d["start"] for d in dn], end=[d["end"] for d in dn], vartype=[0] * len(dn), refs=refs, alts=alts), cutoff)

    groups, cols, first = [], dict(contig=[], rcontig=[], start=[], end=[], vartype=[], refs=[], alts=[]), 0
    want = {}
    for kid in kids:
        dn = [d for d in ds.dnms if d["kid"] == kid]
        c, cutoff = view(dn, kid, cuts[kid])
        dv = abi.dnms_view(cutoff=cutoff, **c)
        r = engine.phase_raw(fams[kid], rhs[kid], dv, P, abi.FIND_SECOND_WINDOW)
        vo, vv = engine.votes(len(dn))
        go, gq = engine.groups(len(dn))
        want[kid] = (r, vo, vv, go, gq)
        groups.append((fams[kid], rhs[kid], first, len(dn), cutoff))
        for k in cols:
            cols[k] += c[k]
        first += len(dn)
    dv_all = abi.dnms_view(cutoff=0.0, **cols)
    for attempt in range(2):  # the second call reuses the merged table
        got = engine.phase_cohort(groups, dv_all, P, want_lists=False)
        vo, vv = engine.votes(first)
        go, gq = engine.groups(first)
        for (fam, rh, f0, cnt, cutoff), kid in zip(groups, kids):
            r, wvo, wvv, wgo, wgq = want[kid]
            for k in ("status", "counts", "origin", "evidence"):
                assert np.array_equal(got[k][f0: f0 + cnt], r[k]), (kid, k)
            for d in range(cnt):
                for j in range(4):
                    assert np.array_equal(vv[vo[4 * (f0 + d) + j]: vo[4 * (f0 + d) + j + 1]], wvv[wvo[4 * d + j]: wvo[4 * d + j + 1]]), (kid, d, j)
                for j in range(2):
                    assert np.array_equal(gq[go[2 * (f0 + d) + j]: go[2 * (f0 + d) + j + 1]], wgq[wgo[2 * d + j]: wgo[2 * d + j + 1]]), (kid, d, j)
    assert sum(int((want[k][0]["status"] == abi.ST_OK).sum()) for k in kids) >= 3
    # a plain uz_phase afterwards is unaffected by the cohort state
    kid = kids[0]
    dn = [d for d in ds.dnms if d["kid"] == kid]
    c, cutoff = view(dn, kid, cuts[kid])
    again = engine.phase_raw(fams[kid], rhs[kid], abi.dnms_view(cutoff=cutoff, **c), P, abi.FIND_SECOND_WINDOW)
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(again[k], want[kid][0][k])
    for kid in kids:
        engine.free_reads(rhs[kid])
    engine.free_sites(sid)


def test_cohort_of_two_hundred_kids_equals_the_oracle_kid_by_kid(engine):
    """The reference's published run is 603 samples in one sites file (README.md:208, unfazed.py:574-575): a cohort batch of 200
    kids x 50 DNMs -- every kid with its own genotype columns, its own alignment records and its own insert cutoff -- through
    ONE uz_phase_cohort, held DNM by DNM against the CPU oracle run kid by kid."""
    import copy
    from oracle import oracle as orc
    from synth import bigsynth
    from synth.sites_np import make_clusters, make_sites, place_dnms_full
    K, D = 200, 50
    from synth.sites_np import DnmColumns
    base = make_sites(600_000, seed=4401, contig_lens=[1.0e8])
    # all DNMs of the cohort at once, on distinct records of the shared table (their REF / ALT are the table's: get_refalt reads the
    # sites file); DNM j belongs to kid j % K
    all_dn = place_dnms_full(base, K * D, seed=5000, indel_frac=0.2)
    P = abi.make_params()
    engine.set_params(P)
    sv = abi.SitesView()
    keep = dict(contig_off=np.ascontiguousarray(base.contig_off, np.int64), pos=base.pos, sflags=base.sflags, ref_base=base.ref_base, alt_base=base.alt_base)
    sv.n_sites, sv.n_contigs = base.n, 1
    for k, a in keep.items():
        setattr(sv, k, a.ctypes.data)
    sites_h = abi.Held(sv, keep)
    sid = engine.upload_sites_view(sites_h)
    groups, want, cols, first = [], [], dict(contig=[], rcontig=[], start=[], end=[], vartype=[], refs=[], alts=[]), 0
    rids = []
    for kid in range(K):
        # the kid's trio: the base genotype columns shifted along the table (every kid sees other genotypes at a site), then its
        # own DNM records made kid het / parents hom-ref with good depth and GQ, as place_dnms_full makes them
        sel = np.arange(kid, K * D, K)
        idx = all_dn.site_idx[sel]
        sc = copy.copy(base)
        sh = 997 * kid + 13
        sc.gt = np.roll(base.gt, sh).copy()
        sc.khap = np.roll(base.khap, sh).copy()
        sc.rd = np.ascontiguousarray(np.roll(base.rd, sh, axis=1))
        sc.ad = np.ascontiguousarray(np.roll(base.ad, sh, axis=1))
        sc.gq = np.ascontiguousarray(np.roll(base.gq, sh, axis=1))
        sc.gt[idx] = 1
        sc.khap[idx] = 0
        for m in range(3):
            sc.rd[m][idx] = 30 if m else 15
            sc.ad[m][idx] = 0 if m else 15
            sc.gq[m][idx] = 99
        dn = DnmColumns(idx, all_dn.contig[sel], all_dn.start[sel], all_dn.end[sel], all_dn.kind[sel], all_dn.length[sel], all_dn.origin[sel],
                        [all_dn.refs[j] for j in sel], [all_dn.alts[j] for j in sel])
        cl = make_clusters(dn)
        cfg = bigsynth.make_cfg(seed=6000 + kid)
        cfg.n_clusters = cl.n
        rh, arrs = bigsynth.reads_cpu(cfg, sc, dn, cl, 0, cl.n, threads=4)
        cutoff = concordant_cutoff(arrs["tlen"][: min(int(rh.view.n_segs), 100000)], P.readlen, 3) + float(kid % 3)
        fam_view = abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)
        dv = abi.dnms_view(dn.contig, dn.contig, dn.start, dn.end, np.zeros(D, np.uint8), dn.refs, dn.alts, cutoff)
        found = orc.find(P, sites_h, fam_view, dv, abi.FIND_SECOND_WINDOW)
        want.append(orc.phase(P, sites_h, rh, dv, found, keep_lists=False))
        fam = engine.add_family(sid, sc.gt, sc.rd, sc.ad, sc.gq)
        packed = io_native.pack_reads(rh, int(P.min_gt_qual), lists=True, with_end=None, cigar_compact=True)
        rid = engine.upload_reads_packed(packed)
        engine.wait_reads(rid)
        rids.append(rid)
        groups.append((fam, rid, first, D, cutoff))
        cols["contig"] += dn.contig.tolist(); cols["rcontig"] += dn.contig.tolist(); cols["start"] += dn.start.tolist(); cols["end"] += dn.end.tolist()
        cols["vartype"] += [0] * D; cols["refs"] += list(dn.refs); cols["alts"] += list(dn.alts)
        first += D
    dv_all = abi.dnms_view(cutoff=0.0, **cols)
    got = engine.phase_cohort(groups, dv_all, P, want_lists=False)
    phased = 0
    for kid in range(K):
        f0 = kid * D
        for k in ("status", "counts", "origin", "evidence"):
            assert np.array_equal(got[k][f0: f0 + D], want[kid][k]), (kid, k)
        phased += int((want[kid]["status"] == abi.ST_OK).sum())
    assert phased > K * D // 10
    for rid in rids:
        engine.free_reads(rid)
    engine.free_sites(sid)
