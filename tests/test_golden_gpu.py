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
