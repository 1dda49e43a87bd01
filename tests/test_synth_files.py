"""The native file writer of the bench generator (synth/uzfiles.cpp): the BAM / BAI and VCF / TBI it writes are held against
the Python writers' indexes byte for byte and decoded back by the product's readers into the tables the generator itself gives."""
import os

import numpy as np
import pytest

from synth import bigsynth
from synth.sites_np import make_clusters, make_sites, place_dnms_full
from unfazed_amd import io_native

import filesio


@pytest.fixture(scope="module")
def workload(tmp_path_factory):
    d = tmp_path_factory.mktemp("uzfiles")
    lens = [3_000_000, 2_000_000, 1_000_000]
    sc = make_sites(12000, seed=77, contig_lens=lens)
    dn = place_dnms_full(sc, 40, seed=78)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=79)
    cfg.n_clusters = cl.n
    bam = str(d / "kid.bam")
    st = bigsynth.write_bam(bam, cfg, sc, dn, cl, contig_len=lens, level=1, threads=3)
    vcf = str(d / "sites.vcf.gz")
    sv = bigsynth.write_vcf(vcf, sc, contig_len=lens, level=1, threads=3)
    return dict(dir=d, lens=lens, sc=sc, dn=dn, cl=cl, cfg=cfg, bam=bam, bam_stats=st, vcf=vcf, vcf_stats=sv)


def test_bam_decodes_to_the_generated_table(workload):
    w = workload
    cl = w["cl"]
    t = io_native.read_bam_table(w["bam"], threads=2)
    held, a = bigsynth.reads_cpu(w["cfg"], w["sc"], w["dn"], cl, 0, cl.n, threads=2)
    n = int(held.view.n_segs)
    assert w["bam_stats"]["records"] == n == t.start.size
    for k in ("start", "end", "flag", "mapq", "tlen", "mate", "n_cigar", "l_seq"):
        assert np.array_equal(getattr(t, k), a[k][:n]), k
    assert np.array_equal(t.cigar, a["cigar"][: int(held.view.n_cigar_total)])
    assert np.array_equal(t.contig_off, a["contig_off"])
    # names: the generator numbers them as a decoder does, in order of first appearance; the writer names global pair p qname_of(p)
    assert np.array_equal(t.qname, a["qname"][:n])
    assert len(set(t.qnames)) == len(t.qnames) == n // 2
    import re
    assert all(re.fullmatch(r"UZSYN:30X:1:\d{3}:\d{7}", t.qnames[i]) for i in (0, 1, len(t.qnames) // 2, len(t.qnames) - 1))
    assert bigsynth.qname_of(998) == "UZSYN:30X:1:001:0000001"
    mate = a["mate"][:n]
    assert np.array_equal(t.qname[mate], t.qname)  # same name <=> same pair
    L = a["l_seq"][:n].astype(np.int64)
    for i in range(0, n, max(1, n // 400)):
        r0, r1 = int(t.sq_off16[i]) << 4, int(a["sq_off16"][i]) << 4
        assert bytes(t.seq[r0: r0 + L[i]]) == bytes(a["seq"][r1: r1 + L[i]])
        assert bytes(t.qual[r0: r0 + L[i]]) == bytes(a["qual"][r1: r1 + L[i]])


def test_bai_equals_the_python_writer(workload, tmp_path):
    w = workload
    ref = filesio.write_bai(w["bam"], str(tmp_path / "ref.bai"))
    assert open(ref, "rb").read() == open(w["bam"] + ".bai", "rb").read()


def test_bam_regions_through_the_native_bai(workload):
    w = workload
    dn = w["dn"]
    tid = dn.contig[::3].astype(np.int32)
    lo = (dn.start[::3] - 1).astype(np.int32)
    hi = (dn.start[::3] + 1).astype(np.int32)
    part = io_native.read_bam_regions(w["bam"], tid, lo, hi, threads=2)
    full = io_native.read_bam_table(w["bam"], threads=2)
    # every record a fetch returns is there, with the same columns as in the whole-file table
    keep = np.zeros(full.start.size, bool)
    tid_of = np.searchsorted(full.contig_off, np.arange(full.start.size), side="right") - 1
    for c, a, b in zip(tid, lo, hi):
        keep |= (tid_of == c) & (full.start < b) & (full.end > a)
    names_full = {full.qnames[int(q)] for q in np.unique(full.qname[keep])}
    names_part = {part.qnames[int(q)] for q in np.unique(part.qname)}
    assert names_full <= names_part


def test_vcf_decodes_to_the_generated_table(workload):
    w = workload
    sc = w["sc"]
    t = io_native.read_vcf_table(w["vcf"], threads=2)
    assert t.pos.size == sc.n == w["vcf_stats"]["records"]
    assert np.array_equal(t.pos, sc.pos)
    assert np.array_equal(t.contig_off, sc.contig_off)
    assert np.array_equal(t.sflags & 1, sc.sflags & 1)
    simple = (sc.sflags & 1) == 0
    assert np.array_equal(t.ref_base[simple], sc.ref_base[simple]) and np.array_equal(t.alt_base[simple], sc.alt_base[simple])
    for m in range(3):
        assert np.array_equal(t.gt[m], (sc.gt >> (2 * m)) & 3)
        for mine, theirs in ((t.ref_depth[m], sc.rd[m]), (t.alt_depth[m], sc.ad[m]), (t.gq[m], sc.gq[m])):
            want = np.where(theirs == 0xFFFF, -1, theirs.astype(np.int64))
            assert np.array_equal(np.asarray(mine).astype(np.int64), want)


def test_tbi_equals_the_python_writer(workload, tmp_path):
    import gzip
    w = workload
    ref = filesio.write_tbi(w["vcf"], str(tmp_path / "ref.tbi"))
    assert gzip.open(ref).read() == gzip.open(w["vcf"] + ".tbi").read()


def test_vcf_regions_through_the_native_tbi(workload):
    w = workload
    sc, dn = w["sc"], w["dn"]
    names = io_native.tabix_contigs(w["vcf"])
    assert names == sc.contig_names[: len(names)]
    ref = dn.contig[::2].astype(np.int32)
    lo = np.maximum(dn.start[::2] - 5000, 0).astype(np.int32)
    hi = (dn.start[::2] + 5000).astype(np.int32)
    part = io_native.read_vcf_table_regions(w["vcf"], ref, lo, hi, threads=2)
    keep = np.zeros(sc.n, bool)
    tid_of = np.searchsorted(sc.contig_off, np.arange(sc.n), side="right") - 1
    for c, a, b in zip(ref, lo, hi):
        keep |= (tid_of == c) & (sc.pos < b) & (sc.pos + np.where((sc.sflags & 1) & (np.arange(sc.n) % 2 == 0), 2, 1) > a)
    assert np.array_equal(part.pos, sc.pos[keep])


def test_filler_between_the_pile_ups_changes_nothing_a_fetch_returns(workload, tmp_path):
    """write_bam(filler=30): read pairs in the gaps between the clusters -- the file is coordinate-sorted, its index reaches every window through
    bins that now hold a lead-in, the walk meets records it has to pass, and a batch's staged records are those of the file without filler."""
    from unfazed_amd import abi
    w = workload
    bam = str(tmp_path / "filled.bam")
    st = bigsynth.write_bam(bam, w["cfg"], w["sc"], w["dn"], w["cl"], contig_len=w["lens"], level=1, threads=3, filler=30.0, filler_reach=40000)
    assert st["records"] > 2 * w["bam_stats"]["records"]
    t = io_native.read_bam_table(bam, threads=2)
    assert t.start.size == st["records"]
    for c in range(len(w["lens"])):  # coordinate-sorted inside every contig
        a, b = int(t.contig_off[c]), int(t.contig_off[c + 1])
        assert (np.diff(t.start[a:b].astype(np.int64)) >= 0).all()
    fill = np.array([nm.startswith("UZFIL:") for nm in t.qnames])
    assert fill.any() and (~fill).sum() == w["bam_stats"]["records"] // 2
    assert np.array_equal(t.qname[t.mate[t.mate >= 0]], t.qname[t.mate >= 0])  # mates name each other
    # the same fetches on both files: the same records (by name, flag and start), however the virtual offsets moved
    dn = w["dn"]
    rng = np.random.default_rng(9)
    c, lo, hi = [], [], []
    for i in range(dn.n):
        c.append(dn.contig[i]); lo.append(dn.start[i] - 1); hi.append(dn.start[i] + 1)
        for p in np.sort(rng.integers(dn.start[i] - 5000, dn.start[i] + 5000, 6)):
            c.append(dn.contig[i]); lo.append(int(p)); hi.append(int(p) + 1)
    fc, flo, fhi = np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32)
    got = []
    for path in (w["bam"], bam):
        src = io_native.BamSource(path, threads=3)
        pk = src.select(fc, flo, fhi, 20)
        n = int(pk.view.n_segs)
        wc = abi.wide_columns(pk)
        names = pk.qnames.take(wc["qname"][:n]) if hasattr(pk.qnames, "take") else [pk.qnames[int(q)] for q in wc["qname"][:n]]
        got.append((n, wc["start"][:n].copy(), wc["tlen"][:n].copy(), wc["mate"][:n].copy(), names, pk.io_stats["records_walked"]))
    assert got[0][0] == got[1][0] > 0
    for k in (1, 2, 3):
        assert np.array_equal(got[0][k], got[1][k])
    assert got[0][4] == got[1][4] and not any(nm.startswith("UZFIL:") for nm in got[1][4])
    assert got[1][5] > 1.5 * got[0][5]  # ... but the walk went past the filler to get there
