"""Cohort batches (uz_phase_cohort): the DNMs of several kids -- own trio columns, own alignment
records, own insert cutoff each -- in one launch sequence give, DNM by DNM, what one uz_phase per kid gives, vote lists
(query-name ids of the kid's own table) included; and the host path, which routes multi-kid batches through it,
reproduces the reference's two-kid goldens (tests/test_golden_gpu.py runs those too)."""
import numpy as np
import pytest

from helpers import tables
from synth.small import SmallConfig, make_small
from unfazed_amd import abi
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
