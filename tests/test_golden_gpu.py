"""The HIP path against the golden vectors the reference itself produced
(tests/golden/*.json), through the C ABI and the host mirrors."""
import glob
import os

import pytest

from test_oracle_golden import GOLD, SNV, check_against_golden, check_cnv_golden, load_snv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", SNV, ids=[os.path.basename(p)[4:-5] for p in SNV])
def test_phase_snvs_golden_gpu(engine, path):
    g, ds = load_snv(path)
    check_against_golden(engine, g, ds)


def test_cnv_allele_balance_golden_gpu(engine):
    check_cnv_golden(engine)


from test_oracle_golden import SV, check_sv_golden  # noqa: E402


@pytest.mark.parametrize("path", SV, ids=[os.path.basename(p)[3:-5] for p in SV])
def test_phase_svs_golden_gpu(engine, path):
    check_sv_golden(engine, path)


# the wide sets (>= 200 DNMs / DEL-DUP / SVs, tie order, autophase): tests/golden/make_golden_wide.py
from test_oracle_golden import (WIDE_SNV, WIDE_SV, check_autophase, check_wide_cnv, check_wide_snv,  # noqa: E402
                                check_wide_sv)


@pytest.mark.parametrize("name", WIDE_SNV, ids=[n[9:-8] for n in WIDE_SNV])
def test_wide_phase_snvs_golden_gpu(engine, name):
    check_wide_snv(engine, name)


def test_wide_cnv_golden_gpu(engine):
    check_wide_cnv(engine)


@pytest.mark.parametrize("name", WIDE_SV, ids=[n[8:-8] for n in WIDE_SV])
def test_wide_phase_svs_golden_gpu(engine, name):
    check_wide_sv(engine, name)


def test_autophase_golden_gpu(engine):
    check_autophase(engine)


from test_oracle_golden import check_find_grid  # noqa: E402


@pytest.mark.parametrize("name", ["find_grid.json", "find_grid_deep.json"])
def test_find_grid_golden_gpu(engine, name):
    """the reference's find() over the genotype x depth grids -- the deep one holds allele depths of 32768, 70000, 10^6 ... that the
    16-bit device columns cannot hold: their classes come from the family's side table (k_site_scan_wide)"""
    check_find_grid(engine, name)
