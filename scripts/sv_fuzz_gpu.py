"""Development aid (GPU box): phase_svs on fresh synth.small_sv seeds, the HIP engine against the oracle backend through the
same host code -- records, record order, per-DNM site lists and stderr must agree.  usage: sv_fuzz_gpu.py FIRST_SEED N"""
import contextlib
import copy
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import RUN_DEFAULTS, dnm_sites, norm_records, tables  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402
from synth.small_sv import SvConfig, make_small_sv  # noqa: E402
from unfazed_amd import session  # noqa: E402
from unfazed_amd.engine import HipEngine  # noqa: E402
from unfazed_amd.sv_phaser import phase_svs  # noqa: E402


def run(backend, ds, sites, reads, name):
    session.set_backend(backend)
    session._HOSTS.clear()
    try:
        session.register_sites(name, sites)
        for k, t in reads.items():
            session.register_reads(k, t)
        a = dict(RUN_DEFAULTS)
        dn = copy.deepcopy(ds.dnms)
        err = io.StringIO()
        with contextlib.redirect_stderr(err):
            recs = phase_svs(dn, list(ds.pedigrees), ds.pedigrees, name, a["threads"], a["build"], a["no_extended"], a["multithread_proc_min"],
                             a["quiet_mode"], a["ab_homref"], a["ab_homalt"], a["ab_het"], a["min_gt_qual"], a["min_depth"], a["search_dist"],
                             a["insert_size_max_sample"], a["stdevs"], a["min_map_qual"], a["readlen"], a["split_error_margin"])
        return recs, dn, err.getvalue()
    finally:
        session.set_backend(None)


first, n = int(sys.argv[1]), int(sys.argv[2])
eng = HipEngine(0)
bad = total = 0
for seed in range(first, first + n):
    ds = make_small_sv(SvConfig(seed=seed))
    sites, reads = tables(ds)
    want, dn_w, err_w = run(OracleBackend(), ds, sites, reads, "mem://svfuzz%d" % seed)
    got, dn_g, err_g = run(eng, ds, sites, reads, "mem://svfuzz%d" % seed)
    ok = norm_records(want) == norm_records(got) and list(want.keys()) == list(got.keys()) and dnm_sites(dn_w) == dnm_sites(dn_g) and err_w == err_g
    total += len(want)
    bad += not ok
    print("sv seed %d: %d SVs, %d records, %s" % (seed, len(ds.dnms), len(want), "ok" if ok else "MISMATCH"), flush=True)
print("sv fuzz: %d seeds, %d records, %d mismatching seeds" % (n, total, bad))
sys.exit(1 if bad else 0)
