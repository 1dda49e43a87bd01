"""CPU-side checks of the C-ABI library: it builds, loads and exports every symbol
include/unfazed_hip.h declares.  No compute call (there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "unfazed_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(uz_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(hip_lib):
    L = ctypes.CDLL(hip_lib)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "missing export " + s


def test_binding_lists_the_same_symbols(hip_lib):
    from unfazed_amd import engine
    assert sorted(engine.EXPORTS) == declared_symbols()
    engine.load_library()


def test_io_library_exports_every_declared_symbol():
    """include/unfazed_io.h (native BAM / VCF decoders) against libunfazed_io.so and its binding."""
    from unfazed_amd import io_native
    txt = open(os.path.join(ROOT, "include", "unfazed_io.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    syms = sorted(set(re.findall(r"\b(uz_[a-z_0-9]+)\s*\(", txt)))
    assert sorted(io_native.IO_EXPORTS) == syms
    L = io_native.load()
    for s in syms:
        assert hasattr(L, s), "missing export " + s


def test_no_device_is_a_loud_error(hip_lib):
    """Without a GPU uz_create must fail with a code, and the Python engine must raise."""
    import torch
    if torch.cuda.is_available():
        return
    from unfazed_amd.engine import HipEngine, UnfazedHipError
    try:
        HipEngine(0)
    except UnfazedHipError:
        return
    raise AssertionError("HipEngine() succeeded without a device")


def test_product_never_imports_the_oracle():
    """The product path must not route through the CPU oracle (or the emulation harness)."""
    import re
    pkg = os.path.join(ROOT, "unfazed_amd")
    bad = re.compile(r"^\s*(from|import)\s+(oracle|emu|oracle_backend)\b|liboracle|libemu_phase|uz_oracle", re.M)
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert not bad.search(src), fn


def test_the_committed_profiles_belong_to_these_device_sources():
    """bench.py quotes a profile's counters (roofline.traffic, issue_model) only while the profile's recorded hash of the device sources equals
    the build's (build.kernel_source_hash): the profiles committed with a round must be the ones of its final kernels."""
    import json
    import os
    from unfazed_amd.build import kernel_source_hash
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = kernel_source_hash()
    for name in ("k1_traffic.json", "phase_issue.json"):
        p = os.path.join(root, "profiles", name)
        assert json.load(open(p)).get("kernel_source_sha") == sha, name
