"""Edge cases of the C ABI on the device: empty batches and tables, DNMs on contigs that the
sites file or the BAM do not have, windows clipped at a contig start, DNMs without any read, call
order errors.  Everything is compared with the oracle where there is something to compare."""
import os

import numpy as np
import pytest

from helpers import tables
from oracle import oracle as orc
from synth.small import SmallConfig, make_small
from unfazed_amd import abi
from unfazed_amd.engine import UnfazedHipError
from unfazed_amd.hostpath import concordant_cutoff
from unfazed_amd.model import ReadsTable, SitesTable

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    ds = make_small(SmallConfig(seed=321, n_dnms=6))
    sites, reads = tables(ds)
    return ds, sites, list(reads.values())[0]


def _dv(sites, rt, dn, contig=None, rcontig=None, start=None):
    refs, alts = [], []
    for d in dn:
        j = int(sites.query(d["chrom"], d["start"], d["start"] + 1)[-1])
        refs.append(sites.ref_str[j].encode())
        alts.append(sites.alt_strs[j][0].encode())
    n = len(dn)
    return abi.dnms_view(
        contig if contig is not None else [sites.contig_index[d["chrom"]] for d in dn],
        rcontig if rcontig is not None else [rt.contig_index[d["chrom"]] for d in dn],
        start if start is not None else [d["start"] for d in dn], [d["end"] for d in dn], [0] * n, refs, alts,
        concordant_cutoff(rt.tlen, 151, 3))


def test_empty_and_missing_inputs(engine, small):
    ds, sites, rt = small
    P = abi.make_params()
    kid = ds.dnms[0]["kid"]
    ped = ds.pedigrees[kid]
    cols = sites.family_columns(kid, ped["dad"], ped["mom"])
    sid = engine.upload_sites(sites)
    fid = engine.add_family(sid, *cols)
    rid = engine.upload_reads(rt)
    sv, fv, rv = abi.sites_view(sites), abi.family_view(*cols), abi.reads_view(rt)
    dn = ds.dnms

    # 1. empty DNM batch
    dv0 = _dv(sites, rt, [])
    r = engine.phase_raw(fid, rid, dv0, P, abi.FIND_SECOND_WINDOW)
    assert r["status"].shape == (0,)
    co, ci, cf, ho, hi = engine.find(fid, dv0, P, abi.FIND_SECOND_WINDOW)
    assert co.tolist() == [0] and ho.tolist() == [0]

    # 2. contig missing from the sites file / from the BAM, window clipped at the contig start
    n = len(dn)
    contig = [sites.contig_index[d["chrom"]] for d in dn]
    rcontig = [rt.contig_index[d["chrom"]] for d in dn]
    start = [d["start"] for d in dn]
    contig[0] = -1
    rcontig[1] = -1
    dv = _dv(sites, rt, dn, contig=contig, rcontig=rcontig, start=start)
    found = orc.find(P, sv, fv, dv, abi.FIND_SECOND_WINDOW)
    want = orc.phase(P, sv, rv, dv, found, keep_lists=False)
    got = engine.phase_raw(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(want[k], got[k]), k
    assert got["status"][0] == abi.ST_NO_CAND

    # 3. a reads table without any record
    empty = ReadsTable.from_segments([], ds.contigs)
    rid0 = engine.upload_reads(empty)
    dvn = _dv(sites, rt, dn)
    got = engine.phase_raw(fid, rid0, dvn, P, abi.FIND_SECOND_WINDOW)
    found = orc.find(P, sv, fv, dvn, abi.FIND_SECOND_WINDOW)
    want = orc.phase(P, sv, abi.reads_view(empty), dvn, found, keep_lists=False)
    assert np.array_equal(want["status"], got["status"])
    assert not (got["status"] == abi.ST_OK).any()
    engine.free_reads(rid0)

    # 4. a sites table without any record
    s0 = SitesTable.from_records([], ds.samples)
    sid0 = engine.upload_sites(s0)
    z = np.zeros((3, 0), np.uint16)
    fid0 = engine.add_family(sid0, np.zeros(0, np.uint8), z, z, z)
    dvz = abi.dnms_view([-1] * n, rcontig, start, [d["end"] for d in dn], [0] * n, [b"A"] * n, [b"C"] * n, 800.0)
    got = engine.phase_raw(fid0, rid, dvz, P, abi.FIND_SECOND_WINDOW)
    assert (got["status"] == abi.ST_NO_CAND).all()
    engine.free_sites(sid0)

    # 5. call-order and handle errors are loud
    with pytest.raises(UnfazedHipError):
        engine.phase_raw(fid, 12345, dvn, P, abi.FIND_SECOND_WINDOW)
    with pytest.raises(UnfazedHipError):
        engine.classify(999, P, 1)
    engine.free_reads(rid)
    engine.free_sites(sid)
    with pytest.raises(UnfazedHipError):
        engine.site_scan(fid)  # the family went away with its sites table


def test_capacity_overflow_is_reported_not_dropped(engine, monkeypatch):
    """A DNM the device layout cannot hold comes back as UZ_ST_CAPACITY: the host prints it (even under --quiet) and
    lists it; it is never a silently missing record.  The overflow is forced by capping the registration scratch."""
    import contextlib
    import copy
    import io
    from helpers import RUN_DEFAULTS, params_from, tables
    from synth.small import SmallConfig, make_small
    from unfazed_amd.hostpath import PhasingHost
    ds = make_small(SmallConfig(seed=4141, n_dnms=24, cluster_prob=0.8))
    sites, reads = tables(ds)
    a = dict(RUN_DEFAULTS, quiet_mode=True)

    def run():
        host = PhasingHost(engine, sites, reads)
        dn = copy.deepcopy(ds.dnms)
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            recs = host.run_read_phasing(dn, ds.pedigrees, 1, "38", False, 1000, True, params_from(a), 5000, 1000000, 3, 151)
        return recs, err.getvalue(), host

    full, err0, _ = run()
    assert err0 == "" and len(full) >= 2
    for cap in (600, 400, 250, 150, 80):  # shrink the registration scratch until some DNM of this dataset no longer fits
        monkeypatch.setenv("UZ_TEST_CAP_T", str(cap))
        part, err1, host = run()
        if host.capacity_skipped:
            break
    monkeypatch.delenv("UZ_TEST_CAP_T")
    assert len(host.capacity_skipped) >= 1
    assert err1.count("UZ_ST_CAPACITY") == len(host.capacity_skipped)
    for k in host.capacity_skipped:
        assert k not in part
    for k, r in part.items():  # the others are untouched
        assert r == full[k]
    assert set(full) - set(part) <= set(host.capacity_skipped)


@pytest.mark.parametrize("arena", [0, 2048, 6000, 12000])
def test_both_builds_of_the_read_stage_agree(engine, monkeypatch, arena):
    """k_phase runs as two kernels: one keeps a DNM's working arrays in its workgroup's LDS arena and gives up the DNMs that do
    not fit, the other redoes those in HBM scratch.  With the arena capped (test hook) every mix of the two must give the
    records of the uncapped run: none, some, most DNMs fitting."""
    import contextlib
    import copy
    import io
    from helpers import RUN_DEFAULTS, dnm_sites, norm_records, params_from, tables
    from synth.small import SmallConfig, make_small
    from unfazed_amd.engine import K_PHASE
    from unfazed_amd.hostpath import PhasingHost
    ds = make_small(SmallConfig(seed=5151, n_dnms=30, cluster_prob=0.5))
    sites, reads = tables(ds)
    a = dict(RUN_DEFAULTS, quiet_mode=True)

    def run():
        host = PhasingHost(engine, sites, reads)
        dn = copy.deepcopy(ds.dnms)
        err = io.StringIO()
        engine.prof_enable(True)
        with contextlib.redirect_stderr(err):
            recs = host.run_read_phasing(dn, ds.pedigrees, 1, "38", False, 1000, True, params_from(a), 5000, 1000000, 3, 151)
        redone = engine.prof_units(K_PHASE)
        engine.prof_enable(False)
        return recs, dn, err.getvalue(), redone

    full, dn0, err0, redone0 = run()
    assert len(full) >= 4
    monkeypatch.setenv("UZ_TEST_PHASE_ARENA", str(arena))
    part, dn1, err1, redone1 = run()
    monkeypatch.delenv("UZ_TEST_PHASE_ARENA")
    assert norm_records(part) == norm_records(full) and list(part) == list(full)
    assert dnm_sites(dn0) == dnm_sites(dn1) and err0 == err1
    assert redone1 >= redone0
    if arena <= 2048:
        assert redone1 >= 1  # (count of the last batch) nothing with candidates fits: redone in HBM scratch


def test_sv_batches_on_a_list_form_table_reproduce_the_goldens(engine, monkeypatch):
    """The list form of the quality plane (counts + the positions of "good" records) serves SV batches too: collect_reads_sv takes
    its evidence under goodread(read, True) from flags, CIGARs and mates alone, and the only quality bits the read stage asks for are
    those of records that pass goodread at het sites.  Forced onto every table of the SV goldens, it must reproduce them -- the
    kernel's guard (a bit of a record without a quality row was asked for) would fail the call loudly otherwise."""
    from test_oracle_golden import SV, WIDE_SV, check_sv_golden, check_wide_sv
    from unfazed_amd import session
    from unfazed_amd.engine import HipEngine
    real = HipEngine.upload_reads

    def forced(self, reads, min_base_qual=None, point_only=False, **kw):
        return real(self, reads, min_base_qual=min_base_qual, point_only=True)

    monkeypatch.setattr(HipEngine, "upload_reads", forced)
    for path in SV:
        session._READS.clear(); session._HOSTS.clear()  # (hosts keep their uploaded tables)
        check_sv_golden(engine, path)
    for name in WIDE_SV:
        session._READS.clear(); session._HOSTS.clear()
        check_wide_sv(engine, name)
    session._READS.clear(); session._HOSTS.clear()


@pytest.mark.parametrize("route", ["walk", "stage", "table"])
def test_sv_goldens_from_indexed_files(engine, tmp_path, monkeypatch, route):
    """The SV goldens from FILES with a BAI next to every BAM: the batch is walked on the device (walk), or travels in the link form of an SV batch -- qualities as lists,
    unit masks from the one-base fetches, no unit for the +-cutoff fetches around the breakpoints -- built in one pass from the file
    (stage) or out of the region table (table: UZ_IO_STAGE=0); the records must be the reference's."""
    import contextlib
    import copy
    import io
    import json
    import sys
    from filesio import dump_dataset, write_bai
    from helpers import RUN_DEFAULTS, norm_records
    from test_oracle_golden import GOLD, SV
    sys.path.insert(0, GOLD)
    from make_golden import dataset_digest
    from synth.small_sv import SvConfig, make_small_sv
    from unfazed_amd import session
    from unfazed_amd.sv_phaser import phase_svs
    if route == "table":
        monkeypatch.setenv("UZ_IO_STAGE", "0")
    # walk: the session's default since round 4 -- blocks inflated, checked and WALKED on the device, the table unpacked from HBM (include/uz_bamwalk.h);
    # stage: the link form built by the host's walk (UZ_WALK=host)
    monkeypatch.setenv("UZ_WALK", "device" if route == "walk" else "host")
    for path in SV:
        g = json.load(open(path))
        ds = make_small_sv(SvConfig(**g["config"]))
        assert dataset_digest(ds) == g["digest"]
        d = tmp_path / (os.path.basename(path)[:-5] + "_" + route)
        paths = dump_dataset(ds, str(d))
        for b in paths["bams"].values():
            write_bai(b)
        session.set_backend(engine)
        session._READS.clear(); session._HOSTS.clear()
        try:
            a = dict(RUN_DEFAULTS)
            a.update(g["run"])
            dn = copy.deepcopy(ds.dnms)
            for x in dn:
                x["bam"] = paths["bams"][x["kid"]]
            err = io.StringIO()
            with contextlib.redirect_stderr(err):
                recs = phase_svs(dn, list(ds.pedigrees), ds.pedigrees, paths["sites"], a["threads"], a["build"], a["no_extended"], a["multithread_proc_min"],
                                 a["quiet_mode"], a["ab_homref"], a["ab_homalt"], a["ab_het"], a["min_gt_qual"], a["min_depth"], a["search_dist"],
                                 a["insert_size_max_sample"], a["stdevs"], a["min_map_qual"], a["readlen"], a["split_error_margin"])
        finally:
            session.set_backend(None)
        assert list(recs.keys()) == g["record_order"]
        assert json.loads(json.dumps(norm_records(recs))) == g["records"]
        assert err.getvalue().splitlines() == g["stderr"]


@pytest.mark.parametrize("shape", [dict(n_dnms=8), dict(readlen=300, ins_mean=800.0, ins_sd=60.0, coverage_per_hap=18.0, n_dnms=6),
                                   dict(readlen=76, ins_mean=250.0, ins_sd=30.0, n_dnms=8, lowq_prob=0.08),
                                   dict(indel_dnm_frac=0.6, indel_prob=0.06, softclip_prob=0.1, n_dnms=8)])
def test_link_form_of_a_point_variant_table_matches_oracle(engine, monkeypatch, shape):
    """The drop-in host uploads whole-file tables with the quality plane (they serve SV batches too); here every table is forced
    into the form a point-variant region table travels in -- two-bit bases, qualities as counts + position lists (two-byte
    positions for reads longer than 256 bases), `end` derived, compact CIGAR -- and the records must equal the oracle's."""
    from helpers import dnm_sites, norm_records, run_host, split_kwargs, tables
    from oracle_backend import OracleBackend
    from synth.small import SmallConfig, make_small
    from unfazed_amd.engine import HipEngine
    real = HipEngine.upload_reads

    def forced(self, reads, min_base_qual=None, point_only=False, **kw):
        return real(self, reads, min_base_qual=min_base_qual, point_only=True)

    monkeypatch.setattr(HipEngine, "upload_reads", forced)
    cfgkw, runkw = split_kwargs(dict(shape))
    if "readlen" in cfgkw:
        runkw["readlen"] = cfgkw["readlen"]
    ds = make_small(SmallConfig(seed=6262, **cfgkw))
    sites, reads = tables(ds)
    want, dn_w, err_w = run_host(OracleBackend(), ds, sites, reads, **runkw)
    got, dn_g, err_g = run_host(engine, ds, sites, reads, **runkw)
    assert dnm_sites(dn_w) == dnm_sites(dn_g)
    assert norm_records(want) == norm_records(got) and list(want.keys()) == list(got.keys())
    assert err_w == err_g and len(want) >= 2


def test_window_lists_are_reused_only_for_the_batch_they_were_made_for(engine, small):
    """The lists of the last finds stay in HBM and a read stage over a batch one of them covered takes them (uz_find_fetch's note in
    unfazed_hip.h).  Whatever changes what a find would give must not meet stale lists: other parameters, another family under a
    recycled id, other DNMs of the same number -- each result is held against the oracle's for ITS inputs."""
    ds, sites, rt = small
    kid = ds.dnms[0]["kid"]
    ped = ds.pedigrees[kid]
    cols = sites.family_columns(kid, ped["dad"], ped["mom"])
    swapped = sites.family_columns(kid, ped["mom"], ped["dad"])  # the parents change places: every candidate's ALT parent flips
    sv, rv = abi.sites_view(sites), abi.reads_view(rt)
    rid = engine.upload_reads(rt)
    dn = ds.dnms

    def check(fid, fam_cols, dv, P):
        fv = abi.family_view(*fam_cols)
        engine.find(fid, dv, P, abi.FIND_SECOND_WINDOW)  # the caller's own find: the read stage below may take its lists
        got = engine.phase_raw(fid, rid, dv, P, abi.FIND_SECOND_WINDOW)
        want = orc.phase(P, sv, rv, dv, orc.find(P, sv, fv, dv, abi.FIND_SECOND_WINDOW), keep_lists=False)
        for k in ("status", "counts", "origin", "evidence"):
            assert np.array_equal(want[k], got[k]), k
        return got

    P = abi.make_params()
    dv = _dv(sites, rt, dn)
    sid = engine.upload_sites(sites)
    fid = engine.add_family(sid, *cols)
    a = check(fid, cols, dv, P)
    assert (a["status"] == abi.ST_OK).any()
    # other parameters: a window that holds nothing but the DNM itself
    P2 = abi.make_params(search_dist=1)
    b = check(fid, cols, dv, P2)
    assert not (b["status"] == abi.ST_OK).any()
    # other DNMs, as many: the first shifted off its site
    dv2 = _dv(sites, rt, dn, start=[dn[0]["start"] + 3000] + [d["start"] for d in dn[1:]])
    check(fid, cols, dv2, P)
    # another family under the same ids
    check(fid, cols, dv, P)
    engine.free_sites(sid)
    sid2 = engine.upload_sites(sites)
    fid2 = engine.add_family(sid2, *swapped)
    assert fid2 == fid  # (the id is recycled: the lists kept for it must be gone)
    c = check(fid2, swapped, dv, P)
    ok = (a["status"] == abi.ST_OK) & (a["origin"] != abi.OR_AMBIGUOUS) & (a["origin"] != abi.OR_NONE)
    assert ok.any() and np.all(c["origin"][ok] != a["origin"][ok])
    engine.free_reads(rid)
    engine.free_sites(sid2)
