"""GPU parity of the whole per-DNM path (K1..K5 behind the C ABI) against the CPU
oracle driven through the same host code, on seeded synthetic trios.  Integer /
byte work: everything must be bit-exact -- records, site lists, messages."""
import numpy as np
import pytest

from helpers import dnm_sites, norm_records, run_host, split_kwargs, tables
from oracle_backend import OracleBackend
from synth.small import SmallConfig, make_small

pytestmark = pytest.mark.gpu

VARIANTS = [
    dict(),
    dict(no_extended=True),
    dict(multithread_proc_min=1),
    dict(chr_prefix="chr"),
    dict(base_err=0.02, cluster_prob=1.0, lowq_prob=0.06),
    dict(min_gt_qual=30, min_depth=0, ab_het=[0.3, 0.7], search_dist=2000),
    dict(kids=["kidA", "kidB"], odd_read_prob=0.15, softclip_prob=0.1, indel_prob=0.08),
    dict(indel_dnm_frac=0.6, mnp_dnm_frac=0.2),
    dict(insert_size_max_sample=20, cluster_prob=1.0),
    dict(readlen=100),
    dict(site_rate=1 / 150.0, cluster_prob=1.0, base_err=0.01),
]


@pytest.mark.parametrize("vi", range(len(VARIANTS)))
def test_records_match_oracle(engine, vi):
    cfgkw, runkw = split_kwargs(VARIANTS[vi])
    ds = make_small(SmallConfig(seed=500 + vi, n_dnms=10, **cfgkw))
    sites, reads = tables(ds)
    want, dn_w, err_w = run_host(OracleBackend(), ds, sites, reads, **runkw)
    got, dn_g, err_g = run_host(engine, ds, sites, reads, **runkw)
    assert dnm_sites(dn_w) == dnm_sites(dn_g)
    assert norm_records(want) == norm_records(got)
    assert list(want.keys()) == list(got.keys())
    assert err_w == err_g
    assert len(want) >= 1
