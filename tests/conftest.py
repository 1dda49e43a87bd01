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
