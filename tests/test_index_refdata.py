"""The native index readers (csrc/io_index.hpp: BAI, TBI) and the tabix-driven region decode, pinned on files HTSLIB wrote: the
indexes and VCFs the reference ships under test/data (copied as data fixtures into tests/golden/refdata)."""
import json
import os

import numpy as np
import pytest

from unfazed_amd import io_native

HERE = os.path.dirname(os.path.abspath(__file__))
RD = os.path.join(HERE, "golden", "refdata")
GOLD = json.load(open(os.path.join(HERE, "golden", "index_refdata.json")))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_index_parse_matches_the_independent_reader(name):
    """committed dump (tests/golden/make_index_golden.py) and the same minimal reader run live, against uz_index_summary"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_index_golden import dump
    live = dump(os.path.join(RD, name))
    assert json.loads(json.dumps(live)) == GOLD[name]
    got = io_native.index_summary(os.path.join(RD, name), GOLD[name]["kind"])
    want = np.array([[r["bins"], r["chunks"], r["linear"], r["sum_beg"], r["sum_end"], r["sum_linear"]] for r in GOLD[name]["refs"]], np.int64)
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    if name == "NA12878.bam.bai":
        assert GOLD[name]["n_ref"] == 87 and any(r["pseudo_bin"] for r in GOLD[name]["refs"])  # the 37450 pseudo-bin is there and left out


@pytest.mark.parametrize("name", ["trio_hets_snvs_chr22.vcf.gz", "trio_hets_svs_chr22.vcf.gz", "trio_svs_chr22.vcf.gz"])
def test_region_decode_through_the_real_tbi_equals_whole_file_decode_restricted(name):
    path = os.path.join(RD, name)
    full = io_native.read_vcf_table(path, threads=2)
    names = io_native.tabix_contigs(path)
    assert names == [c for c in full.contigs if full.contig_off[full.contig_index[c] + 1] > full.contig_off[full.contig_index[c]]]
    rng = np.random.default_rng(len(name))
    pos, end = np.asarray(full.pos, np.int64), np.asarray(full.end, np.int64)
    lo_all, hi_all = int(pos.min()), int(end.max())
    for trial in range(200):
        k = int(rng.integers(1, 6))
        width = int(rng.choice([1, 10, 1000, 20000, 300000]))
        lo = rng.integers(max(0, lo_all - 5000), hi_all + 5000, k)
        hi = lo + rng.integers(1, width + 1, k)
        ref = np.zeros(k, np.int32)
        part = io_native.read_vcf_table_regions(path, ref, lo, hi, threads=2)
        keep = np.zeros(pos.size, bool)
        for a, b in zip(lo, hi):
            keep |= (pos < b) & (end > a)
        assert np.array_equal(part.pos, full.pos[keep]), (trial, lo, hi)
        assert np.array_equal(part.gt, full.gt[:, keep])
        assert np.array_equal(part.ref_depth, full.ref_depth[:, keep]) and np.array_equal(part.alt_depth, full.alt_depth[:, keep])
        idx = np.nonzero(keep)[0]
        for j in range(0, idx.size, max(1, idx.size // 5)):
            assert part.lines[j] == full.lines[int(idx[j])]
