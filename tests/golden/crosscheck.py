"""Live cross-check of the CPU oracle against the REFERENCE itself on FRESH seeds (not the committed fixtures).
Authoring container only (needs /root/reference; imported through tests/refshim):

    python tests/golden/crosscheck.py --seeds 30 [--sv-seeds 9] [--first-seed 9000]

For every seed a synth.small trio is generated, phased by the imported reference (phase_snvs / phase_svs) and by the
host path over the oracle backend; records, record order, per-DNM site lists and stderr must be identical.
Exit code 0 and a one-line summary when everything matches."""
import argparse
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import refrun  # noqa: E402
from helpers import norm_records, run_host  # noqa: E402
from oracle_backend import OracleBackend  # noqa: E402
from synth.small import SmallConfig, make_small  # noqa: E402


def variants(rng):
    """a configuration + run options drawn per seed: the knobs the golden sets pin one at a time, mixed"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=30)
    ap.add_argument("--sv-seeds", type=int, default=9)
    ap.add_argument("--first-seed", type=int, default=9000)
    args = ap.parse_args()
    assert refrun.available(), "/root/reference is required"
    n_dnm = n_rec = bad = 0
    for k in range(args.seeds):
        seed = args.first_seed + k
        cfg, run = variants(np.random.RandomState(seed))
        ds = make_small(SmallConfig(seed=seed, **cfg))
        want, wdn, werr, _ = refrun.run_phase_snvs(ds, tag="x%d" % seed, **run)
        got, gdn, gerr = run_host(OracleBackend(), ds, **run)
        ok = (list(want) == list(got) and json.loads(json.dumps(norm_records(want))) == json.loads(json.dumps(norm_records(got)))
              and werr.splitlines() == gerr.splitlines()
              and [(d.get("candidate_sites"), d.get("het_sites")) for d in wdn] == [(d.get("candidate_sites"), d.get("het_sites")) for d in gdn])
        n_dnm += len(ds.dnms)
        n_rec += len(want)
        bad += not ok
        print("snv seed %d: %d DNMs, %d records, %s  %s %s" % (seed, len(ds.dnms), len(want), "ok" if ok else "MISMATCH", cfg, run), flush=True)
    from synth.small_sv import SvConfig, make_small_sv
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import _phase_svs_through
    for k in range(args.sv_seeds):
        seed = args.first_seed + 500 + k
        ds = make_small_sv(SvConfig(seed=seed, n_svs=6))
        run = {"no_extended": True} if k % 3 == 2 else {}
        want, wdn, werr, _ = refrun.run_phase_svs(ds, tag="xs%d" % seed, **run)
        got, gdn, gerr = _phase_svs_through(OracleBackend(), ds, run)
        ok = (list(want) == list(got) and json.loads(json.dumps(norm_records(want))) == json.loads(json.dumps(norm_records(got)))
              and werr.splitlines() == gerr.splitlines())
        n_rec += len(want)
        bad += not ok
        print("sv seed %d: %d SVs, %d records, %s" % (seed, len(ds.dnms), len(want), "ok" if ok else "MISMATCH"), flush=True)
    print("crosscheck: %d snv seeds (%d DNMs), %d sv seeds, %d records, %d mismatching runs" % (args.seeds, n_dnm, args.sv_seeds, n_rec, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
