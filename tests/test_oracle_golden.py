"""The CPU oracle (and the host logic around it) against the golden vectors produced
by the reference itself (tests/golden/make_golden.py).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

from helpers import norm_records, run_host, tables
from oracle import oracle as orc
from oracle_backend import OracleBackend
from synth.small import SmallConfig, make_small
from unfazed_amd import abi, summarize
from unfazed_amd.hostpath import PhasingHost, concordant_cutoff
from unfazed_amd.model import SiteRecord, SitesTable

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SNV = sorted(glob.glob(os.path.join(GOLD, "snv_*.json")))


def load_snv(path):
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import dataset_digest
    g = json.load(open(path))
    ds = make_small(SmallConfig(**g["config"]))
    assert dataset_digest(ds) == g["digest"], "synthetic generator drifted from the fixture inputs"
    return g, ds


def check_against_golden(backend, g, ds):
    recs, dnms, err = run_host(backend, ds, **g["run"])
    assert list(recs.keys()) == g["record_order"]
    assert json.loads(json.dumps(norm_records(recs))) == g["records"]
    ref = {(d["chrom"], d["start"], d["end"], d["kid"]): d for d in g["dnms"]}
    for d in dnms:
        r = ref[(d["chrom"], d["start"], d["end"], d["kid"])]
        assert d.get("candidate_sites") == r.get("candidate_sites")
        assert d.get("het_sites") == r.get("het_sites")
    assert [d["start"] for d in dnms] is not None
    assert err.splitlines() == g["stderr"]


@pytest.mark.parametrize("path", SNV, ids=[os.path.basename(p)[4:-5] for p in SNV])
def test_phase_snvs_golden(path):
    g, ds = load_snv(path)
    check_against_golden(OracleBackend(), g, ds)


def check_find_grid(backend, name):
    """find_grid_deep: every record holds an allele depth beyond 16 bits (32768, 70000, 10^6 ...) in at least one member -- the
    reference takes any depth (informative_site_finder.py:46-73); such sites travel on the side table of the family view"""
    g = json.load(open(os.path.join(GOLD, name)))
    recs = [SiteRecord("1", s["start"], s["ref"], s["alts"], s["gt"], s["rd"], s["ad"], s["gq"]) for s in g["sites"]]
    sites = SitesTable.from_records(recs, g["samples"])
    ped = {"kid": {"kid": "kid", "dad": "dad", "mom": "mom", "sex": "2"}}
    host = PhasingHost(backend, sites, {})
    n_c = 0
    for c in g["cases"]:
        ps = c["params"]
        P = abi.make_params(min_gt_qual=ps["min_gt_qual"], min_depth=ps["min_depth"], ab_homref=ps["ab_homref"],
                            ab_homalt=ps["ab_homalt"], ab_het=ps["ab_het"])
        dn = [dict(chrom="1", start=c["dnm"]["start"], end=c["dnm"]["end"], kid="kid", vartype=c["dnm"]["vartype"],
                   bam="", cram_ref=None)]
        host.find(dn, ped, c["search_dist"], 1, "38", 1000, True, P, whole_region=c["whole_region"])
        assert dn[0]["candidate_sites"] == c["candidate_sites"]
        assert dn[0]["het_sites"] == c["het_sites"]
        n_c += len(c["candidate_sites"])
    assert n_c > 150
    if "deep" in name:
        assert sites.wide_depths is not None and len(sites.wide_depths[0]) >= 600


@pytest.mark.parametrize("name", ["find_grid.json", "find_grid_deep.json"])
def test_find_grid_golden(name):
    check_find_grid(OracleBackend(), name)


def test_binary_search_golden():
    for c in json.load(open(os.path.join(GOLD, "bsearch.json"))):
        assert orc.bsearch(c["start"], c["end"], c["pos"]) == c["order"], c


def test_summarize_golden():
    g = json.load(open(os.path.join(GOLD, "summarize.json")))
    for c in g["cases"]:
        got = summarize.summarize_record(c["record"], c["include_ambiguous"], True, c["ratio"])
        assert got == c["summary"], c
    for b in g["beds"]:
        assert summarize.bed_lines(b["records"], b["include_ambiguous"], b["verbose"], 10) == b["lines"]


def test_cutoff_golden():
    for c in json.load(open(os.path.join(GOLD, "cutoff.json"))):
        head = np.array(c["tlen"][: c["insert_size_max_sample"] + 1], dtype=np.int32)
        assert float(concordant_cutoff(head, c["readlen"], 3)) == c["cutoff"]
        assert orc.concordant_cutoff(head, c["readlen"]) == c["cutoff"]


def run_cnv(backend, ds, ps):
    import contextlib
    import copy
    import io
    sites, reads = tables(ds)
    host = PhasingHost(backend, sites, reads)
    P = abi.make_params(min_gt_qual=ps["min_gt_qual"], min_depth=ps["min_depth"], ab_homref=ps["ab_homref"],
                        ab_homalt=ps["ab_homalt"], ab_het=ps["ab_het"])
    dn = copy.deepcopy(ds.dnms)
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        recs = host.run_cnv_phasing(dn, ds.pedigrees, 1, "38", 1000, False, P)
    return recs, dn, err.getvalue()


def check_cnv_golden(backend):
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import dataset_digest, make_cnv_dataset
    g = json.load(open(os.path.join(GOLD, "cnv.json")))
    ds = make_cnv_dataset()
    assert dataset_digest(ds) == g["digest"]
    for c in g["cases"]:
        recs, dn, err = run_cnv(backend, ds, c["params"])
        assert list(recs.keys()) == c["record_order"]
        assert json.loads(json.dumps(recs)) == c["records"]
        assert err.splitlines() == c["stderr"]
        for d, r in zip(dn, c["dnms"]):
            assert d.get("candidate_sites") == r.get("candidate_sites")
            assert d.get("het_sites") == r.get("het_sites")
        for k, r in recs.items():
            assert summarize.summarize_record(r, True, True, 10) == c["summaries"][k]


def test_cnv_allele_balance_golden():
    check_cnv_golden(OracleBackend())


SV = sorted(glob.glob(os.path.join(GOLD, "sv_*.json")))


def check_sv_golden(backend, path):
    import contextlib
    import copy
    import io
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import dataset_digest
    from helpers import RUN_DEFAULTS
    from synth.small_sv import SvConfig, make_small_sv
    from unfazed_amd import session
    from unfazed_amd.sv_phaser import phase_svs
    g = json.load(open(path))
    ds = make_small_sv(SvConfig(**g["config"]))
    assert dataset_digest(ds) == g["digest"]
    sites, reads = tables(ds)
    session.set_backend(backend)
    try:
        session.register_sites("mem://svsites", sites)
        for k, t in reads.items():
            session.register_reads(k, t)
        a = dict(RUN_DEFAULTS)
        a.update(g["run"])
        dn = copy.deepcopy(ds.dnms)
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            recs = phase_svs(dn, list(ds.pedigrees), ds.pedigrees, "mem://svsites", a["threads"], a["build"],
                             a["no_extended"], a["multithread_proc_min"], a["quiet_mode"], a["ab_homref"], a["ab_homalt"],
                             a["ab_het"], a["min_gt_qual"], a["min_depth"], a["search_dist"], a["insert_size_max_sample"],
                             a["stdevs"], a["min_map_qual"], a["readlen"], a["split_error_margin"])
    finally:
        session.set_backend(None)
    assert list(recs.keys()) == g["record_order"]
    assert json.loads(json.dumps(norm_records(recs))) == g["records"]
    assert err.getvalue().splitlines() == g["stderr"]
    for d, r in zip(dn, g["dnms"]):
        assert d.get("candidate_sites") == r.get("candidate_sites")
        assert d.get("het_sites") == r.get("het_sites")
    for k, r in recs.items():
        assert summarize.summarize_record(r, True, False, 10) == g["summaries"][k]


@pytest.mark.parametrize("path", SV, ids=[os.path.basename(p)[3:-5] for p in SV])
def test_phase_svs_golden(path):
    check_sv_golden(OracleBackend(), path)


def test_oracle_count_decision_is_summarize_record():
    """The integer decision the oracle (and K6 on the device) makes from the four read-backed and two CNV counts is
    summarize_record's (the host restatement pinned by the 1160 reference cases of summarize.json), `ambig` quirk included."""
    rng = np.random.default_rng(3)
    vals = [0, 0, 0, 1, 2, 3, 9, 10, 11, 20, 25, 100]
    rows = []
    for _ in range(4000):
        dr, mr, cd, cm = (int(rng.choice(vals)) for _ in range(4))
        ds = int(rng.integers(0, 4)) if dr else 0
        ms = int(rng.integers(0, 4)) if mr else 0
        rows.append((dr, mr, ds, ms, cd, cm))
    rows = np.array(rows, np.int32)
    for ratio in (10, 1, 2):
        org, ev, et = orc.summarize_counts(rows[:, :4], rows[:, 4:], ratio)
        for k, (dr, mr, ds, ms, cd, cm) in enumerate(rows.tolist()):
            rec = dict(region=dict(chrom="1", start=5, end=9), vartype="DEL", kid="k", dad="D", mom="M", evidence_type="readbacked",
                       dad_reads=["r%d" % i for i in range(dr)], mom_reads=["q%d" % i for i in range(mr)],
                       dad_sites=[str(i) for i in range(ds)], mom_sites=[str(100 + i) for i in range(ms)],
                       cnv_dad_sites=[str(200 + i) for i in range(cd)], cnv_mom_sites=[str(300 + i) for i in range(cm)], cnv_evidence_type="")
            full = summarize.summarize_record(rec, True, False, ratio)
            strict = summarize.summarize_record(rec, False, False, ratio)
            want_org = {None: abi.OR_NONE, "D": abi.OR_DAD, "M": abi.OR_MOM, "D|M": abi.OR_AMBIGUOUS}[full["origin_parent"]]
            assert org[k] == want_org and ev[k] == full["evidence_count"], (rows[k], ratio)
            assert [nm for bit, nm in abi.ET_NAMES if et[k] & bit] == full["evidence_types"], (rows[k], ratio, et[k])
            dropped = bool(et[k] & abi.ET_AMBIG_FLAG) or org[k] == abi.OR_NONE
            assert (strict is None) == dropped, (rows[k], ratio)


# ------------------------------------------------------------------ the wide sets (tests/golden/make_golden_wide.py)
def _load_gz(name):
    import gzip
    return json.loads(gzip.open(os.path.join(GOLD, name), "rb").read().decode())


WIDE_SNV = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "wide_snv_*.json.gz")))
WIDE_SV = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLD, "wide_sv_*.json.gz")))


def _digest_of(ds):
    import sys
    sys.path.insert(0, GOLD)
    from make_golden import dataset_digest
    return dataset_digest(ds)


def check_wide_snv(backend, name):
    from helpers import compact_dnms, compact_records, reverse_ties
    g = _load_gz(name)
    ds = make_small(SmallConfig(**g["config"]))
    if g["reverse_ties"]:
        reverse_ties(ds)
    assert _digest_of(ds) == g["digest"], "synthetic generator drifted from the fixture inputs"
    recs, dnms, err = run_host(backend, ds, **g["run"])
    assert list(recs.keys()) == g["record_order"]
    assert json.loads(json.dumps(compact_records(recs))) == g["records"]
    assert json.loads(json.dumps(compact_dnms(dnms))) == g["dnms"]
    assert err.splitlines() == g["stderr"]
    return g, recs


@pytest.mark.parametrize("name", WIDE_SNV, ids=[n[9:-8] for n in WIDE_SNV])
def test_wide_phase_snvs_golden(name):
    g, recs = check_wide_snv(OracleBackend(), name)
    if "ties" not in name:
        assert len(g["dnms"]) >= 200


def test_tie_order_is_part_of_the_input():
    """connect_reads is first-come: the same records with the ties of the coordinate order reversed give the reference
    (and therefore the goldens) different haplotype groups for some DNMs -- the two fixtures must differ, and each is
    reproduced exactly above."""
    a, b = _load_gz("wide_snv_ties_forward.json.gz"), _load_gz("wide_snv_ties_reversed.json.gz")
    assert a["record_order"] == b["record_order"]
    assert a["records"] != b["records"]


def check_wide_cnv(backend):
    import sys
    sys.path.insert(0, GOLD)
    from helpers import compact_dnms
    from make_golden import make_cnv_dataset
    g = _load_gz("wide_cnv.json.gz")
    ds = make_cnv_dataset(seed=g["seed"], n=g["n"])
    assert _digest_of(ds) == g["digest"]
    assert sum(d["vartype"] in ("DEL", "DUP") for d in ds.dnms) >= 160
    for c in g["cases"]:
        recs, dn, err = run_cnv(backend, ds, c["params"])
        assert list(recs.keys()) == c["record_order"]
        assert json.loads(json.dumps(recs)) == c["records"]
        assert err.splitlines() == c["stderr"]
        assert json.loads(json.dumps(compact_dnms(dn))) == c["dnms"]
        for k, r in recs.items():
            assert summarize.summarize_record(r, True, True, 10) == c["summaries"][k]


def test_wide_cnv_golden():
    check_wide_cnv(OracleBackend())


def _phase_svs_through(backend, ds, run):
    import contextlib
    import copy
    import io
    from helpers import RUN_DEFAULTS
    from unfazed_amd import session
    from unfazed_amd.sv_phaser import phase_svs
    sites, reads = tables(ds)
    session.set_backend(backend)
    try:
        session.register_sites("mem://svsites", sites)
        for k, t in reads.items():
            session.register_reads(k, t)
        a = dict(RUN_DEFAULTS)
        a.update(run)
        dn = copy.deepcopy(ds.dnms)
        for d in dn:
            d["bam"] = "mem://%s.bam" % d["kid"]
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            recs = phase_svs(dn, list(ds.pedigrees), ds.pedigrees, "mem://svsites", a["threads"], a["build"],
                             a["no_extended"], a["multithread_proc_min"], a["quiet_mode"], a["ab_homref"], a["ab_homalt"],
                             a["ab_het"], a["min_gt_qual"], a["min_depth"], a["search_dist"], a["insert_size_max_sample"],
                             a["stdevs"], a["min_map_qual"], a["readlen"], a["split_error_margin"])
    finally:
        session.set_backend(None)
    return recs, dn, err.getvalue()


def check_wide_sv(backend, name):
    from helpers import compact_dnms, compact_records
    from synth.small_sv import SvConfig, make_small_sv
    g = _load_gz(name)
    ds = make_small_sv(SvConfig(**g["config"]))
    assert _digest_of(ds) == g["digest"]
    recs, dn, err = _phase_svs_through(backend, ds, g["run"])
    assert list(recs.keys()) == g["record_order"]
    assert json.loads(json.dumps(compact_records(recs))) == g["records"]
    assert err.splitlines() == g["stderr"]
    assert json.loads(json.dumps(compact_dnms(dn))) == g["dnms"]
    for k, r in recs.items():
        assert summarize.summarize_record(r, True, False, 10) == g["summaries"][k]


@pytest.mark.parametrize("name", WIDE_SV, ids=[n[8:-8] for n in WIDE_SV])
def test_wide_phase_svs_golden(name):
    check_wide_sv(OracleBackend(), name)


def _norm_summary(s):
    """verbose read-name lists come out in Python set order (quirk Q19): compare them sorted"""
    if s is None:
        return None
    s = dict(s)
    for k in ("origin_parent_reads", "other_parent_reads"):
        if k in s and s[k] != "-":
            s[k] = ",".join(sorted(s[k].split(",")))
    return s


def check_autophase(backend):
    """chrX / chrY DNMs x kid sex x build 37 / 38 / na x every PAR edge, through phase_snvs (per-DNM find and
    find_many) and phase_svs (quirk Q18: the SV driver falls through after writing the SEX-CHROM record)."""
    import sys
    sys.path.insert(0, GOLD)
    from helpers import compact_dnms, compact_records
    from make_golden_wide import autophase_dataset
    g = json.load(open(os.path.join(GOLD, "autophase.json")))
    n_auto = 0
    for run in g["runs"]:
        ds = autophase_dataset(run["prefix"])
        if run["kind"] == "sv":
            for i, d in enumerate(ds.dnms):
                d["vartype"] = ["DEL", "DUP", "INV"][i % 3]
                d["end"] = d["start"] + 700 + 50 * (i % 7)
        assert _digest_of(ds) == run["digest"]
        if run["kind"] == "snv":
            recs, dn, err = run_host(backend, ds, **run["run"])
        else:
            recs, dn, err = _phase_svs_through(backend, ds, run["run"])
        assert list(recs.keys()) == run["record_order"], run["run"]
        assert json.loads(json.dumps(compact_records(recs))) == run["records"]
        assert err.splitlines() == run["stderr"]
        assert json.loads(json.dumps(compact_dnms(dn))) == run["dnms"]
        for k, r in recs.items():
            assert _norm_summary(summarize.summarize_record(r, True, True, 10)) == _norm_summary(run["summaries"][k])
        n_auto += sum(1 for r in recs.values() if r["evidence_type"].startswith("SEX-CHROM"))
    assert n_auto > 150


def test_autophase_golden():
    check_autophase(OracleBackend())
