"""The whole driver (file decoders -> phase_snvs -> BED / annotated VCF text) against what the
reference's own driver produced on the same files (tests/golden/cli.json).  The CPU test drives the
host code through the oracle backend; the -m gpu test runs `python -m unfazed_amd` on the device."""
import contextlib
import io
import json
import os
import subprocess
import sys

import pytest

from filesio import dump_dataset
from synth.small import SmallConfig, make_small
from test_oracle_golden import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(tmp_path):
    sys.path.insert(0, GOLD)
    from make_golden import dataset_digest
    g = json.load(open(os.path.join(GOLD, "cli.json")))
    ds = make_small(SmallConfig(**g["config"]))
    assert dataset_digest(ds) == g["digest"]
    return g, ds, dump_dataset(ds, str(tmp_path))


def _argv(paths, run):
    a = ["-d", paths[run["dnms"]], "-s", paths["sites"], "-p", paths["ped"], "--build", "38", "-t", "1", "-q",
         "-o", run["output_type"]]
    if run["include_ambiguous"]:
        a.append("--include-ambiguous")
    if run["verbose"]:
        a.append("--verbose")
    return a + ["--bam-pairs"] + ["%s:%s" % (k, v) for k, v in paths["bams"].items()]


def _check(run, text, n_samples):
    lines = text.splitlines()
    if run["output_type"] == "bed":
        if not run["verbose"]:
            assert lines == run["lines"]
            return
        assert len(lines) == len(run["lines"]) and lines[0] == run["lines"][0]
        for got, want in zip(lines[1:], run["lines"][1:]):
            g, w = got.split("\t"), want.split("\t")
            assert g[:10] == w[:10] and g[11] == w[11]
            # read names come out of a Python set in the reference: order is hash-dependent (quirk Q19)
            assert sorted(g[10].split(",")) == sorted(w[10].split(","))
            assert sorted(g[12].split(",")) == sorted(w[12].split(","))
        return
    body = [l for l in lines if not l.startswith("#")]
    head = [l for l in lines if l.startswith("##")]
    assert any(l.startswith("##unfazed=1.0.3. Phase info in pipe-separated GT field order") for l in head)
    assert sum(l.startswith("##FORMAT=<ID=UOPS,Number=1,Type=Float") for l in head) == 1
    assert sum(l.startswith("##FORMAT=<ID=UET,Number=1,Type=Float") for l in head) == 1
    assert len(body) == len(run["body"])
    for l, w in zip(body, run["body"]):
        f = l.split("\t")
        assert f[0] == w["chrom"] and int(f[1]) == w["pos"] and f[8].endswith(":UOPS:UET")
        for i in range(n_samples):
            col = f[9 + i].split(":")
            a0, a1, ph = w["genotypes"][i]
            want_gt = ("%s%s%s" % ("." if a0 < 0 else a0, "|" if ph else "/", "." if a1 < 0 else a1))
            assert col[0] == want_gt, (l, i)
            assert float(col[-2]) == w["uops"][i] and float(col[-1]) == w["uet"][i]


def _index_inputs(paths):
    """a BAI next to every BAM, and the sites VCF rewritten as BGZF with a tabix index next to it"""
    import gzip
    from filesio import write_bai, write_bgzf_text, write_tbi
    for b in paths["bams"].values():
        write_bai(b)
    text = gzip.open(paths["sites"], "rt").read()
    write_bgzf_text(paths["sites"], text)
    write_tbi(paths["sites"])


@pytest.mark.parametrize("indexed", [False, True], ids=["whole_file", "bai_tbi_regions"])
def test_cli_matches_reference_driver(tmp_path, indexed):
    """indexed: a BAI sits next to every BAM and a tabix index next to the sites VCF, so the session decodes only the regions
    each batch looks at (uz_bam_decode_regions, uz_vcf_decode_regions) instead of the whole files -- the output must not change."""
    from oracle_backend import OracleBackend
    from unfazed_amd import session
    from unfazed_amd.__main__ import setup_args
    from unfazed_amd.unfazed import unfazed
    g, ds, paths = _inputs(tmp_path)
    if indexed:
        _index_inputs(paths)
    session.set_backend(OracleBackend())
    session._READS.clear()
    session._HOSTS.clear()
    try:
        for run in g["runs"]:
            args = setup_args().parse_args(_argv(paths, run))
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                unfazed(args)
            _check(run, buf.getvalue(), len(ds.samples))
        # (the sites really came through the tabix index: region tables are cached under NAME@batch)
        assert any("@" in k for k in session._SITES) == indexed
    finally:
        session.set_backend(None)
        for k in [k for k in session._SITES if "@" in k]:
            del session._SITES[k]


@pytest.mark.gpu
@pytest.mark.parametrize("indexed", [False, True], ids=["whole_file", "bai_tbi_regions"])
def test_cli_on_device_matches_reference_driver(tmp_path, hip_lib, indexed):
    g, ds, paths = _inputs(tmp_path)
    if indexed:
        _index_inputs(paths)
    for run in g["runs"]:
        out = subprocess.run([sys.executable, "-m", "unfazed_amd"] + _argv(paths, run), cwd=ROOT, check=True,
                             capture_output=True, text=True)
        _check(run, out.stdout, len(ds.samples))
