import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a gfx950 (MI355X) device")


@pytest.fixture(scope="session")
def hip_lib():
    from unfazed_amd import build
    return build.build()


@pytest.fixture(scope="session")
def engine(hip_lib):
    from unfazed_amd.engine import HipEngine
    e = HipEngine(0)
    yield e
    e.close()


@pytest.fixture(scope="session")
def workload(tmp_path_factory):
    """a synthetic coordinate-sorted BAM + BAI over three contigs with 60 DNM pile-ups (tests/test_io_stage.py, test_stage_desc.py, test_bamwalk_gpu.py)"""
    from synth import bigsynth
    from synth.sites_np import make_clusters, make_sites, place_dnms_full
    d = tmp_path_factory.mktemp("stage")
    lens = [3_000_000, 2_000_000, 1_000_000]
    sc = make_sites(12000, seed=177, contig_lens=lens)
    dn = place_dnms_full(sc, 60, seed=178)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=179)
    cfg.n_clusters = cl.n
    bam = str(d / "kid.bam")
    bigsynth.write_bam(bam, cfg, sc, dn, cl, contig_len=lens, level=1, threads=3)
    return dict(sc=sc, dn=dn, cl=cl, bam=bam)
