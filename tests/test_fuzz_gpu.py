"""Randomised differential test of the whole path (HIP through the C ABI vs the CPU oracle through the
same host code) on shapes the fixed variants do not reach: deep coverage (pair tables beyond the
register-sort and LDS-arena sizes, BFS levels with hundreds of winners), very dense het sites, short
reads, tiny windows.  Bit-exact: records, per-DNM site lists, messages, haplotype groups."""
import os

import numpy as np
import pytest

from helpers import dnm_sites, norm_records, run_host, split_kwargs, tables
from oracle_backend import OracleBackend
from synth.small import SmallConfig, make_small

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(coverage_per_hap=45.0, n_dnms=5),                                   # ~90x: M > 1024
    dict(coverage_per_hap=30.0, site_rate=1 / 80.0, cluster_prob=1.0, n_dnms=5),   # dense het sites, M > 512
    dict(coverage_per_hap=22.0, base_err=0.03, lowq_prob=0.05, n_dnms=6),     # noisy: many failed claims, more levels
    dict(readlen=76, ins_mean=250.0, ins_sd=30.0, coverage_per_hap=20.0, n_dnms=6),
    dict(search_dist=400, site_rate=1 / 60.0, n_dnms=8),                     # tiny windows
    dict(coverage_per_hap=4.0, n_dnms=10),                                   # thin coverage: many empty outcomes
    dict(coverage_per_hap=35.0, indel_dnm_frac=0.5, indel_prob=0.05, softclip_prob=0.08, n_dnms=5),
    dict(coverage_per_hap=28.0, no_extended=True, n_dnms=6),
    dict(insert_size_max_sample=6, n_dnms=6),   # doubles as the enumerate cut-off of the het-site fetch loop (:178-179): it bites everywhere
]


@pytest.mark.parametrize("si", range(len(SHAPES)))
def test_random_shapes_match_oracle(engine, si):
    kw = dict(SHAPES[si])
    cfgkw, runkw = split_kwargs(kw)
    for k in ("search_dist", "readlen"):  # both a property of the data and a run parameter
        if k in cfgkw:
            runkw[k] = cfgkw[k]
    ds = make_small(SmallConfig(seed=int(os.environ.get("UZ_FUZZ_SEED", "9000")) + 17 * si, **cfgkw))
    sites, reads = tables(ds)
    want, dn_w, err_w = run_host(OracleBackend(), ds, sites, reads, **runkw)
    got, dn_g, err_g = run_host(engine, ds, sites, reads, **runkw)
    assert dnm_sites(dn_w) == dnm_sites(dn_g)
    assert norm_records(want) == norm_records(got)
    assert list(want.keys()) == list(got.keys())
    assert err_w == err_g
