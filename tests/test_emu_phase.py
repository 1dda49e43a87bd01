"""The read-stage KERNEL BODY (unfazed_amd/csrc/phase_body.hpp), compiled for the CPU as
a one-lane emulation, against the oracle.  This pins the kernel's logic in the authoring
container (no GPU); the real kernel is checked by the -m gpu tests."""
import numpy as np
import pytest

from emu import emu
from oracle import oracle as orc
from synth.small import SmallConfig, make_small
from unfazed_amd import abi
from unfazed_amd.hostpath import concordant_cutoff
from unfazed_amd.model import ReadsTable, SitesTable

CASES = [
    dict(),
    dict(no_extended=True),
    dict(base_err=0.02, cluster_prob=1.0, lowq_prob=0.06),
    dict(kids=["kidA", "kidB"], odd_read_prob=0.15, softclip_prob=0.1, indel_prob=0.08, indel_dnm_frac=0.5),
    dict(coverage_per_hap=45.0, n_dnms=3),                                          # pair tables beyond 1024 entries
    dict(coverage_per_hap=30.0, site_rate=1 / 80.0, cluster_prob=1.0, n_dnms=3),    # dense het sites
    dict(coverage_per_hap=22.0, base_err=0.03, lowq_prob=0.05, n_dnms=4),            # noisy
    dict(read_goal=7),       # the enumerate cut-off of the fetch loop (:178-179) bites at every het site
    dict(read_goal=0, coverage_per_hap=8.0),
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_kernel_body_matches_oracle(ci):
    kw = dict(CASES[ci])
    no_ext = kw.pop("no_extended", False)
    read_goal = kw.pop("read_goal", None)
    ds = make_small(SmallConfig(seed=700 + ci, **dict(dict(n_dnms=8), **kw)))
    sites = SitesTable.from_records(ds.sites, ds.samples)
    P = abi.make_params(no_extended=no_ext)
    if read_goal is not None:
        P.read_goal = read_goal
    sv = abi.sites_view(sites)
    n_ok = 0
    for kid in ds.reads:
        rt = ReadsTable.from_segments(ds.reads[kid], ds.contigs)
        ped = ds.pedigrees[kid]
        fv = abi.family_view(*sites.family_columns(kid, ped["dad"], ped["mom"]))
        rv = abi.reads_view(rt)
        dn = [d for d in ds.dnms if d["kid"] == kid]
        refs, alts = [], []
        for d in dn:
            j = int(sites.query(d["chrom"], d["start"], d["start"] + 1)[-1])
            refs.append(sites.ref_str[j].encode())
            alts.append(sites.alt_strs[j][0].encode())
        dv = abi.dnms_view([sites.contig_index[d["chrom"]] for d in dn], [rt.contig_index[d["chrom"]] for d in dn],
                           [d["start"] for d in dn], [d["end"] for d in dn], [0] * len(dn), refs, alts,
                           concordant_cutoff(rt.tlen, P.readlen, 3))
        found = orc.find(P, sv, fv, dv, abi.FIND_SECOND_WINDOW)
        want = orc.phase(P, sv, rv, dv, found, keep_lists=True)
        got = emu.phase(P, sv, rv, dv, found)
        # the twin holds the qualities as the staged form does: a bit of a record with more than 10 low-quality bases (never
        # "good") must never be asked for
        assert got["base_err"] == 0
        for k in ("status", "counts", "origin", "evidence"):
            assert np.array_equal(want[k], got[k]), k
        vo, vv, go, gq = want["vote_off"], want["vote_val"], want["grp_off"], want["grp_q"]
        for d in range(len(dn)):
            for j in range(4):
                assert np.array_equal(vv[vo[4 * d + j]: vo[4 * d + j + 1]], got["lists"][d][j])
            if not no_ext and want["status"][d] == abi.ST_OK:
                for j in range(2):
                    assert np.array_equal(gq[go[2 * d + j]: go[2 * d + j + 1]], got["lists"][d][4 + j])
        n_ok += int((want["status"] == abi.ST_OK).sum())
    assert n_ok >= 1


def test_kernel_body_matches_oracle_sv():
    """SV read-backed path: collect_reads_sv around both breakpoints, then the shared chaining / vote."""
    from synth.small_sv import SvConfig, make_small_sv
    from unfazed_amd.hostpath import vartype_code
    ds = make_small_sv(SvConfig(seed=91, n_svs=8))
    sites = SitesTable.from_records(ds.sites, ds.samples)
    P = abi.make_params()
    sv = abi.sites_view(sites)
    kid = "kid1"
    rt = ReadsTable.from_segments(ds.reads[kid], ds.contigs)
    ped = ds.pedigrees[kid]
    fv = abi.family_view(*sites.family_columns(kid, ped["dad"], ped["mom"]))
    rv = abi.reads_view(rt)
    dn = ds.dnms
    n = len(dn)
    dv = abi.dnms_view([sites.contig_index[d["chrom"]] for d in dn], [rt.contig_index[d["chrom"]] for d in dn],
                       [d["start"] for d in dn], [d["end"] for d in dn], [vartype_code(d["vartype"]) for d in dn],
                       [b""] * n, [b""] * n, concordant_cutoff(rt.tlen, P.readlen, 3))
    found = orc.find(P, sv, fv, dv, abi.FIND_SECOND_WINDOW)
    want = orc.phase(P, sv, rv, dv, found, keep_lists=True)
    got = emu.phase(P, sv, rv, dv, found)
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(want[k], got[k]), k
    vo, vv = want["vote_off"], want["vote_val"]
    for d in range(n):
        for j in range(4):
            assert np.array_equal(vv[vo[4 * d + j]: vo[4 * d + j + 1]], got["lists"][d][j])
    assert int((want["status"] == abi.ST_OK).sum()) >= 2
