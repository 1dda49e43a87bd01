"""The input readers on the reference's own small test files (tests/golden/refdata/, copies of
/root/reference/test/data): the variants, pedigree entries and messages must equal
what the reference's read_vars_bed / read_vars_vcf / parse_ped gave (tests/golden/refdata_plumbing.json,
produced by tests/golden/make_refdata_golden.py), for the text decoder and for the native one; and the BED
writer's header is the one of the reference's expected output file.  CPU only."""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from unfazed_amd import io_native, unfazed as drv
from unfazed_amd.io_vcf import read_vcf
from unfazed_amd.model import SitesTable
from unfazed_amd.summarize import write_bed_output

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "golden", "refdata")
G = json.load(open(os.path.join(HERE, "golden", "refdata_plumbing.json")))


@pytest.mark.parametrize("name", sorted(G["bed"]))
def test_bed_reader(name):
    assert list(drv.read_vars_bed(os.path.join(D, name))) == G["bed"][name]


@pytest.mark.parametrize("name", sorted(G["vcf"]))
def test_vcf_reader(name):
    got = list(drv.read_vars_vcf(os.path.join(D, name)))
    assert got == G["vcf"][name]
    assert len(got) > 50


@pytest.mark.parametrize("key", sorted(G["ped"]))
def test_ped_parser(key):
    ped = key.split("|")[0]
    want = G["ped"][key]
    err = io.StringIO()
    drv.QUIET_MODE = False
    with contextlib.redirect_stderr(err):
        got = drv.parse_ped(os.path.join(D, ped), set(want["kids"]))
    assert got == want["entries"]
    assert sorted(err.getvalue().splitlines()) == want["stderr"]


@pytest.mark.parametrize("name", sorted(G["vcf"]))
def test_native_decoder_on_the_reference_vcfs(name):
    path = os.path.join(D, name)
    samples, recs, _ = read_vcf(path)
    want = SitesTable.from_records(recs, samples)
    got = io_native.read_vcf_table(path, threads=2)
    assert got.samples == want.samples == ["NA12878", "NA12891", "NA12892"]
    for c in ("contig_off", "pos", "end", "sflags", "ref_base", "alt_base", "gt", "ref_depth", "alt_depth"):
        assert np.array_equal(getattr(got, c), getattr(want, c)), c
    assert np.array_equal(got.gq, want.gq)


def test_bed_header_is_the_reference_header(tmp_path):
    out = str(tmp_path / "o.bed")
    write_bed_output({}, False, False, out, 10)
    ours = open(out).read().splitlines()[0]
    theirs = open(os.path.join(D, "trio_hets_snvs_chr22_phased.bed")).read().splitlines()[0]
    assert ours == theirs
