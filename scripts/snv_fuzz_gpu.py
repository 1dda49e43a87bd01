"""Development aid (GPU box): the SNV / indel read stage on fresh synth.small seeds with the knobs of the golden sets mixed per
seed (two kids, noisy clustered reads, odd reads, indel / MNP DNMs, chr prefix, --no-extended, find_many, thresholds) -- the HIP
engine against the oracle backend through the same host code.  usage: snv_fuzz_gpu.py FIRST_SEED N"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from helpers import dnm_sites, norm_records, run_host, tables  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402
from synth.small import SmallConfig, make_small  # noqa: E402
from unfazed_amd.engine import HipEngine  # noqa: E402


def variants(rng):
    cfg = dict(n_dnms=int(rng.randint(6, 14)))
    run = {}
    if rng.rand() < 0.3:
        cfg["kids"] = ["kidA", "kidB"]
    if rng.rand() < 0.3:
        cfg.update(base_err=0.02, cluster_prob=1.0, lowq_prob=0.05)
    if rng.rand() < 0.3:
        cfg.update(odd_read_prob=0.12, softclip_prob=0.08, indel_prob=0.06)
    if rng.rand() < 0.3:
        cfg.update(indel_dnm_frac=0.5, mnp_dnm_frac=0.2)
    if rng.rand() < 0.2:
        cfg["chr_prefix"] = "chr"
    if rng.rand() < 0.25:
        run["no_extended"] = True
    if rng.rand() < 0.25:
        run["multithread_proc_min"] = 1
    if rng.rand() < 0.2:
        run.update(min_gt_qual=int(rng.choice([10, 30])), search_dist=int(rng.choice([2000, 8000])))
    return cfg, run


first, n = int(sys.argv[1]), int(sys.argv[2])
eng = HipEngine(0)
bad = total = dnms = 0
for seed in range(first, first + n):
    cfg, run = variants(np.random.RandomState(seed))
    ds = make_small(SmallConfig(seed=seed, **cfg))
    sites, reads = tables(ds)
    want, dn_w, err_w = run_host(OracleBackend(), ds, sites, reads, **run)
    got, dn_g, err_g = run_host(eng, ds, sites, reads, **run)
    ok = norm_records(want) == norm_records(got) and list(want.keys()) == list(got.keys()) and dnm_sites(dn_w) == dnm_sites(dn_g) and err_w == err_g
    total += len(want)
    dnms += len(ds.dnms)
    bad += not ok
    if not ok or seed % 20 == 0:
        print("snv seed %d: %d DNMs, %d records, %s %s %s" % (seed, len(ds.dnms), len(want), "ok" if ok else "MISMATCH", cfg, run), flush=True)
print("snv fuzz: %d seeds, %d DNMs, %d records, %d mismatching seeds" % (n, dnms, total, bad))
sys.exit(1 if bad else 0)
