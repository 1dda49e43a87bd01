"""Native decoders (libunfazed_io.so, include/unfazed_io.h) against the Python decoders that state
the formats (unfazed_amd/io_bam.py, io_vcf.py): every column of the two tables must be equal, on
the synthetic trio files, on hand-made edge cases, and on the VCFs the reference ships under
test/data when this container has them.  CPU only."""
import ctypes as C
import gzip
import os
import struct

import numpy as np
import pytest

from synth.small import SmallConfig, make_small
from synth.small_sv import SvConfig, make_small_sv
from tests.filesio import dump_dataset
from unfazed_amd import io_native
from unfazed_amd.io_bam import _bgzf_block, read_bam, write_bam
from unfazed_amd.io_vcf import read_vcf
from unfazed_amd.model import (FMREVERSE, FPAIRED, FPROPER, FREAD1, FREAD2, FREVERSE, FSUPP, FUNMAP, OP_D, OP_I, OP_M, OP_S,
                               ReadsTable, Segment, SitesTable)

READ_COLS = ["contig_off", "max_span", "start", "end", "flag", "mapq", "aux", "tlen", "qname", "mate", "cigar_off",
             "n_cigar", "cigar", "l_seq", "sq_off16", "seq", "qual"]
SITE_COLS = ["contig_off", "pos", "end", "sflags", "ref_base", "alt_base", "gt", "ref_depth", "alt_depth"]


def _same_reads(path, threads):
    contigs, segs = read_bam(path)
    want = ReadsTable.from_segments(segs, contigs)
    got = io_native.read_bam_table(path, threads=threads, insert_size_max_sample=50)
    assert got.contigs == want.contigs
    for c in READ_COLS:
        a, b = getattr(got, c), getattr(want, c)
        assert a.dtype == b.dtype, c
        assert np.array_equal(a, b), c
    assert list(got.qnames) == list(want.qnames)
    assert np.array_equal(got.tlen_head, np.array([s.tlen for s in segs[:51]], dtype=np.int32))
    return got


def _same_sites(path, threads):
    samples, recs, header = read_vcf(path)
    want = SitesTable.from_records(recs, samples)
    got = io_native.read_vcf_table(path, threads=threads)
    assert got.samples == want.samples and got.contigs == want.contigs
    for c in SITE_COLS:
        a, b = getattr(got, c), getattr(want, c)
        assert a.shape == b.shape, c
        assert np.array_equal(a, b), c
    assert np.array_equal(got.gq, want.gq, equal_nan=True)
    assert list(got.ref_str) == list(want.ref_str)
    assert list(got.alt_strs) == list(want.alt_strs)
    assert got.header == header
    assert [ln.split("\t") for ln in got.lines] == [r.raw for r in recs]
    return got


def test_exports():
    lib = io_native.load()
    for name in io_native.IO_EXPORTS:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_synthetic_trio_files(tmp_path, threads):
    ds = make_small(SmallConfig(seed=5, n_dnms=12))
    paths = dump_dataset(ds, str(tmp_path))
    for bam in paths["bams"].values():
        t = _same_reads(bam, threads)
        assert t.n_segs > 1000
    _same_sites(paths["sites"], threads)
    _same_sites(paths["dnm_vcf"], threads)


def test_sv_reads_with_sa_tags_and_supplementary(tmp_path):
    ds = make_small_sv(SvConfig(seed=3, n_svs=4))
    paths = dump_dataset(ds, str(tmp_path))
    for bam in paths["bams"].values():
        t = _same_reads(bam, 4)
        assert (t.aux & 2).any()  # SA tags seen


def _seg(name, flag, tid, pos, cigar, mtid, mpos, tlen, seq, qual, mapq=60, sa=False):
    return Segment(name, flag, tid, pos, mapq, cigar, mtid, mpos, tlen, seq, qual, has_sa=sa)


def test_bam_edge_cases(tmp_path):
    q = [30] * 10
    segs = [
        _seg("a", FPAIRED | FREAD1 | FMREVERSE, 0, 100, [(OP_M, 10)], 0, 300, 210, "ACGTACGTAC", q),
        _seg("dup", FPAIRED | FREAD1, 0, 100, [(OP_S, 2), (OP_M, 5), (OP_I, 1), (OP_M, 2)], 0, 100, 0, "ACGTACGTAC", q),
        _seg("dup", FPAIRED | FREAD2, 0, 100, [(OP_M, 4), (OP_D, 3), (OP_M, 6)], 0, 100, 0, "ACGTACGTAN", q),
        _seg("noqual", FPAIRED | FREAD1, 0, 120, [(OP_M, 10)], 0, 500, 0, "ACGTACGTAC", None),  # 0xFF qualities
        _seg("noseq", 0, 0, 130, [(OP_M, 10)], -1, -1, 0, "", None),
        _seg("nocigar", FUNMAP | FPAIRED | FREAD2, 0, 140, [], 0, 140, 0, "ACGTA", [20] * 5),
        _seg("a", FPAIRED | FREAD2 | FREVERSE, 0, 300, [(OP_M, 10)], 0, 100, -210, "TTTTTTTTTT", q),
        _seg("a", FPAIRED | FREAD2 | FSUPP, 0, 305, [(OP_S, 5), (OP_M, 5)], 0, 100, -210, "TTTTTGGGGG", q, sa=True),
        _seg("odd", FPAIRED | FREAD1, 1, 5, [(OP_M, 7)], 0, 100, 0, "ACGTACG", [1, 2, 3, 4, 5, 6, 7]),  # mate on another contig
        _seg("solo", 0, 2, 9, [(OP_M, 3)], -1, -1, 0, "AAA", [9, 9, 9]),
        _seg("unplaced", FUNMAP, -1, -1, [], -1, -1, 0, "ACGT", [5, 5, 5, 5]),  # dropped from the table, kept in tlen_head
    ]
    path = str(tmp_path / "edge.bam")
    write_bam(path, [("1", 1000), ("2", 1000), ("3", 1000), ("empty", 50)], segs)
    t = _same_reads(path, 2)
    assert t.n_segs == 10 and len(t.qnames) == 7
    assert t.mate[0] == 6 and t.mate[6] == 0  # pysam mate(): first record of the name that fits


def test_bam_errors(tmp_path):
    lib = io_native.load()
    with pytest.raises(io_native.IoError) as e:
        io_native.read_bam_table(str(tmp_path / "missing.bam"))
    assert e.value.code == -1
    bad = tmp_path / "bad.bam"
    bad.write_bytes(_bgzf_block(b"NOTABAMFILE....") + _bgzf_block(b""))
    with pytest.raises(io_native.IoError) as e:
        io_native.read_bam_table(str(bad))
    assert e.value.code == -2
    q = [30] * 4
    unsorted = [_seg("x", 0, 0, 50, [(OP_M, 4)], -1, -1, 0, "ACGT", q), _seg("y", 0, 0, 10, [(OP_M, 4)], -1, -1, 0, "ACGT", q)]
    p2 = str(tmp_path / "unsorted.bam")
    write_bam(p2, [("1", 1000)], unsorted)
    with pytest.raises(io_native.IoError) as e:
        io_native.read_bam_table(p2)
    assert e.value.code == -3
    with pytest.raises(ValueError):
        contigs, segs = read_bam(p2)
        ReadsTable.from_segments(segs, contigs)
    # a flipped payload byte fails the block CRC
    data = bytearray(open(p2, "rb").read())
    data[40] ^= 0x55
    p3 = tmp_path / "crc.bam"
    p3.write_bytes(bytes(data))
    with pytest.raises(io_native.IoError) as e:
        io_native.read_bam_table(str(p3))
    assert e.value.code == -2
    assert lib.uz_io_last_error()


VCF_EDGE = """##fileformat=VCFv4.2
##source=edge cases
#CHROM	POS	ID	REF	ALT	QUAL	FILTER	INFO	FORMAT	kid	dad	mom
1	10	.	A	G	.	.	.	GT:AD:GQ	0/1:10,12:99	0|0:20,0:45.5	1/1:0,31:.
1	10	.	A	G,T	.	.	X=1;END=55;Y	GT:AD:GQ	1/2:1,2,3:7	./.:.:.	0/0:9:3
1	11	.	AT	A	.	.	END=abc	GT:GQ:AD	0/1:50:5,6	.:.:.	./1:12:.,4
1	11	.	C	*	.	.	SVTYPE=DEL	GT:RO:AO:GQ	0/1:7:8,9:20	1:3:4:1e1	0:.:.:nan
1	12	.	C	.	.	.	.	GT	0/0	0/1
1	13	.	G	<DEL>	.	.	SVTYPE=DEL;END=400	GT:AD	0/1	1/.	2/2:1,2

2	5	.	T	C	.	.	.	AD:GT:GQ:AD	1,1:0/1:5:30,40	.,.:0/0:-1:.	7:1|1:0:.
2	5	.	T	C
"""


def test_vcf_edge_cases(tmp_path):
    for name, opener in (("edge.vcf", open), ("edge.vcf.gz", gzip.open)):
        p = str(tmp_path / name)
        with opener(p, "wt") as fh:
            fh.write(VCF_EDGE)
        t = _same_sites(p, 2)
        assert t.n_sites == 8 and t.contigs == ["1", "2"]
    # bgzip framing of the same text (several blocks)
    raw = VCF_EDGE.encode()
    p = tmp_path / "edge.bgz.vcf.gz"
    p.write_bytes(b"".join(_bgzf_block(raw[i:i + 97]) for i in range(0, len(raw), 97)) + _bgzf_block(b""))
    _same_sites(str(p), 3)


def test_vcf_errors(tmp_path):
    p = tmp_path / "ungrouped.vcf"
    p.write_text("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n1\t5\t.\tA\tC\t.\t.\t.\n2\t5\t.\tA\tC\t.\t.\t.\n1\t9\t.\tA\tC\t.\t.\t.\n")
    with pytest.raises(io_native.IoError) as e:
        io_native.read_vcf_table(str(p))
    assert e.value.code == -3
    p2 = tmp_path / "unsorted.vcf"
    p2.write_text("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n1\t9\t.\tA\tC\t.\t.\t.\n1\t5\t.\tA\tC\t.\t.\t.\n")
    with pytest.raises(io_native.IoError) as e:
        io_native.read_vcf_table(str(p2))
    assert e.value.code == -3


REF_DATA = "/root/reference/test/data"


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference's test data is only present in the authoring container")
@pytest.mark.parametrize("name", ["trio_hets_snvs_chr22.vcf.gz", "trio_hets_svs_chr22.vcf.gz", "trio_svs_chr22.vcf.gz"])
def test_reference_vcfs(name):
    path = os.path.join(REF_DATA, name)
    if not os.path.exists(path):
        pytest.skip("absent")
    t = _same_sites(path, 4)
    assert t.n_sites > 0


def test_family_columns_agree(tmp_path):
    ds = make_small(SmallConfig(seed=9, n_dnms=6))
    paths = dump_dataset(ds, str(tmp_path))
    samples, recs, _ = read_vcf(paths["sites"])
    want = SitesTable.from_records(recs, samples)
    got = io_native.read_vcf_table(paths["sites"], threads=2)
    kid = list(ds.pedigrees)[0]
    p = ds.pedigrees[kid]
    for a, b in zip(got.family_columns(kid, p["dad"], p["mom"]), want.family_columns(kid, p["dad"], p["mom"])):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("use_idx,int16", [(False, False), (True, True)])
def test_bcf_matches_the_text_form(tmp_path, use_idx, int16):
    """BCF2 input (the reference opens .bcf through cyvcf2, utils.py:7): the native decoder must give the same
    columns for the binary form as for the text form of the same records."""
    import gzip as gz
    from tests.bcfio import write_bcf
    from tests.filesio import vcf_text
    ds = make_small(SmallConfig(seed=21, n_dnms=8))
    sv = make_small_sv(SvConfig(seed=4, n_svs=3))
    for k, d in enumerate((ds, sv)):
        recs = list(d.sites)
        for j, r in enumerate(recs[:20]):  # give some records INFO fields
            if j % 3 == 0:
                r.info = {"SVTYPE": "DEL", "END": r.start + 50}
                r.end = r.start + 50
        text_path = str(tmp_path / ("t%d.vcf.gz" % k))
        with gz.open(text_path, "wt") as fh:
            fh.write(vcf_text(d.samples, recs, d.contigs))
        bcf_path = str(tmp_path / ("t%d.bcf" % k))
        write_bcf(bcf_path, d.samples, recs, d.contigs, use_idx=use_idx, int16_depths=int16)
        a = io_native.read_vcf_table(text_path, threads=2)
        b = io_native.read_vcf_table(bcf_path, threads=3)
        assert b.is_bcf and not a.is_bcf
        assert a.samples == b.samples and a.contigs == b.contigs
        for c in SITE_COLS:
            assert np.array_equal(getattr(a, c), getattr(b, c)), c
        assert np.array_equal(a.gq, b.gq)
        assert list(a.ref_str) == list(b.ref_str)
        assert list(a.alt_strs) == list(b.alt_strs)
        for j in range(min(25, a.n_sites)):
            assert a.info(j, "SVTYPE") == b.info(j, "SVTYPE")
        assert a.info(0, "SVTYPE") == "DEL" and a.info(1, "SVTYPE") is None


def test_bcf_dnm_input_gives_the_same_variants(tmp_path):
    from tests.bcfio import write_bcf
    from unfazed_amd.unfazed import read_vars_vcf
    ds = make_small(SmallConfig(seed=3, n_dnms=6, kids=["kidA", "kidB"]))
    paths = dump_dataset(ds, str(tmp_path))
    smp, recs, _ = read_vcf(paths["dnm_vcf"])
    bcf = str(tmp_path / "dnms.bcf")
    write_bcf(bcf, smp, recs, ds.contigs)
    a, b = list(read_vars_vcf(paths["dnm_vcf"])), list(read_vars_vcf(bcf))
    assert len(a) >= 6 and a == b


# ---------------------------------------------------------------------------- index-driven region decode
def _restrict_whole(t, tid, lo, hi):
    """indices of the records of a whole-file table the fetches return, closed under mate()"""
    n = t.n_segs
    contig_of = np.searchsorted(t.contig_off, np.arange(n), side="right") - 1
    keep = np.zeros(n, bool)
    for c, a, b in zip(tid, lo, hi):
        keep |= (contig_of == c) & (t.start < b) & (t.end > a)
    for _ in range(6):
        m = t.mate[keep]
        before = keep.sum()
        keep[m[m >= 0]] = True
        if keep.sum() == before:
            break
    return np.nonzero(keep)[0]


def test_region_decode_equals_whole_file_decode_restricted(tmp_path):
    """uz_bam_decode_regions through a BAI: same records, columns, names and mate links as the whole-file decode
    restricted to what the fetches return + mates -- and only a fraction of the file is inflated."""
    from filesio import dump_dataset, write_bai
    from synth.small import SmallConfig, make_small
    ds = make_small(SmallConfig(seed=515, n_dnms=14, kids=["kidA"], odd_read_prob=0.2, softclip_prob=0.05, indel_prob=0.05))
    paths = dump_dataset(ds, str(tmp_path))
    bam = paths["bams"]["kidA"]
    write_bai(bam)
    whole = io_native.read_bam_table(bam)
    rng = np.random.default_rng(5)
    for trial in range(4):
        # fetch points around a few DNMs: single bases and a wider window, on both contigs
        pick = rng.choice(len(ds.dnms), size=3 + trial, replace=False)
        tid, lo, hi = [], [], []
        for k in pick:
            d = ds.dnms[k]
            c = whole.contig_index[d["chrom"]]
            for off in (0, -700, 1300, 2500):
                tid.append(c); lo.append(d["start"] + off); hi.append(d["start"] + off + 1)
            tid.append(c); lo.append(d["start"] - 1500); hi.append(d["start"] - 1100)
        part = io_native.read_bam_regions(bam, tid, lo, hi)
        idx = _restrict_whole(whole, tid, lo, hi)
        assert part.n_segs == idx.size and idx.size > 50
        for col in ("start", "end", "tlen", "flag", "mapq", "aux", "n_cigar", "l_seq"):
            assert np.array_equal(getattr(part, col), getattr(whole, col)[idx]), col
        # names: ids are interned per table, the strings must agree record by record
        assert [part.qnames[int(q)] for q in part.qname] == [whole.qnames[int(q)] for q in whole.qname[idx]]
        new = np.full(whole.n_segs, -1, np.int64)
        new[idx] = np.arange(idx.size)
        wm = whole.mate[idx]
        assert np.array_equal(part.mate, np.where(wm >= 0, new[np.maximum(wm, 0)], -1))
        for i in rng.integers(0, idx.size, 60):
            j = int(idx[i])
            assert np.array_equal(part.cigar[part.cigar_off[i]: part.cigar_off[i] + part.n_cigar[i]],
                                  whole.cigar[whole.cigar_off[j]: whole.cigar_off[j] + whole.n_cigar[j]])
            ls = int(part.l_seq[i])
            assert np.array_equal(part.seq[int(part.sq_off16[i]) * 16: int(part.sq_off16[i]) * 16 + ls],
                                  whole.seq[int(whole.sq_off16[j]) * 16: int(whole.sq_off16[j]) * 16 + ls])
            assert np.array_equal(part.qual[int(part.sq_off16[i]) * 16: int(part.sq_off16[i]) * 16 + ls],
                                  whole.qual[int(whole.sq_off16[j]) * 16: int(whole.sq_off16[j]) * 16 + ls])
        assert np.array_equal(part.contig_off, np.searchsorted(idx, whole.contig_off))
        assert np.array_equal(part.tlen_head, whole.tlen_head)
        st = part.io_stats
        assert st["records_kept"] == idx.size
        assert st["records_walked"] < 0.7 * whole.n_segs  # (head excluded: it is read for the insert-size estimate)
    # no fetch at all: an empty table with the header and the head
    empty = io_native.read_bam_regions(bam, [], [], [])
    assert empty.n_segs == 0 and empty.contigs == whole.contigs


def _big_vcf_text(n_per_contig=30000, contigs=("chr1", "chr2", "chrX"), seed=3):
    """position-sorted sites on several contigs, with a few long deletions and INFO/END records reaching into later windows"""
    rng = np.random.default_rng(seed)
    lines = ["##fileformat=VCFv4.2"] + ["##contig=<ID=%s>" % c for c in contigs] + [
        '##INFO=<ID=END,Number=1,Type=Integer,Description="end">', '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
        '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths">', '##FORMAT=<ID=GQ,Number=1,Type=Float,Description="Genotype quality">',
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tkid\tdad\tmom"]
    gts = ["0/0", "0/1", "1/1", "./."]
    for c in contigs:
        pos = np.cumsum(rng.integers(1, 400, n_per_contig)) + 1000
        for k, p in enumerate(pos.tolist()):
            ref, alt, info = "ACGT"[k % 4], "CGTA"[k % 4], "."
            if k % 997 == 0:
                ref = "A" * int(rng.integers(50, 3000))  # a long deletion
                alt = "A"
            elif k % 1013 == 0:
                info = "SVTYPE=DEL;END=%d" % (p + int(rng.integers(1000, 20000)))
                alt = "<DEL>"
            cols = ["%s:%d,%d:%d" % (gts[int(g)], int(a), int(b), int(q)) for g, a, b, q in
                    zip(rng.integers(0, 4, 3), rng.integers(0, 40, 3), rng.integers(0, 40, 3), rng.integers(0, 99, 3))]
            lines.append("\t".join([c, str(p), ".", ref, alt, "50", "PASS", info, "GT:AD:GQ"] + cols))
    return "\n".join(lines) + "\n"


def test_vcf_region_decode_through_the_tabix_index(tmp_path):
    """uz_vcf_decode_regions: the header and the records overlapping the intervals, read through NAME.tbi -- the same columns,
    strings and raw lines as the whole-file decode restricted to those records, from a fraction of the file"""
    from filesio import write_bgzf_text, write_tbi
    path = os.path.join(str(tmp_path), "sites.vcf.gz")
    write_bgzf_text(path, _big_vcf_text(), block_bytes=20000)
    write_tbi(path)
    assert io_native.tabix_index_path(path) == path + ".tbi"
    names = io_native.tabix_contigs(path)
    assert names == ["chr1", "chr2", "chrX"]
    whole = io_native.read_vcf_table(path)
    rng = np.random.default_rng(8)
    pick = np.sort(rng.choice(whole.pos.size, 40, replace=False))
    ref = np.searchsorted(whole.contig_off, pick, "right") - 1
    lo, hi = whole.pos[pick] - 5002, whole.pos[pick] + 5002
    t = io_native.read_vcf_table_regions(path, [names.index(whole.contigs[r]) for r in ref], lo, hi)
    keep = np.zeros(whole.pos.size, bool)
    for r, a, b in zip(ref, lo, hi):
        c0, c1 = int(whole.contig_off[r]), int(whole.contig_off[r + 1])
        keep[c0:c1] |= (whole.pos[c0:c1] < b) & (whole.end[c0:c1] > a)
    idx = np.nonzero(keep)[0]
    assert 1000 < idx.size < 0.2 * whole.pos.size and t.pos.size == idx.size
    assert (whole.end[idx] - whole.pos[idx]).max() > 1000  # long records reaching into a window came along
    for k in ("pos", "end", "sflags", "ref_base", "alt_base"):
        assert np.array_equal(getattr(whole, k)[idx], getattr(t, k)), k
    for k in ("gt", "ref_depth", "alt_depth", "gq"):
        assert np.array_equal(getattr(whole, k)[:, idx], getattr(t, k)), k
    assert t.samples == whole.samples and t.header == whole.header
    assert [c for c in whole.contigs if c in t.contigs] == t.contigs
    for j in range(0, idx.size, 37):
        i = int(idx[j])
        assert whole.lines[i] == t.lines[j] and whole.ref_str[i] == t.ref_str[j] and whole.alt_strs[i] == t.alt_strs[j]
    # only a fraction of the file was read and inflated
    assert t.io_stats[0] < 0.35 * os.path.getsize(path) and t.io_stats[3] == idx.size and t.io_stats[2] < 0.35 * whole.pos.size
    # no interval, an interval past the end of a contig, a reference the index does not have
    empty = io_native.read_vcf_table_regions(path, [], [], [])
    assert empty.pos.size == 0 and empty.samples == whole.samples
    far = io_native.read_vcf_table_regions(path, [2], [2_000_000_000 - 10], [2_000_000_000])
    assert far.pos.size == 0
    with pytest.raises(io_native.IoError):
        io_native.read_vcf_table_regions(path, [7], [0], [10])
    with pytest.raises(io_native.IoError):
        io_native.read_vcf_table_regions(path, [0], [0], [10], tbi=path)  # not a tabix index


def test_session_asks_the_tabix_index_for_the_batch_windows(tmp_path, monkeypatch):
    """session.site_regions: DNM windows (+- search_dist, whole events) in the index's own sequence names -- the chr prefix is
    decided by the FIRST sequence of the index (= the first record of the file, utils.py:46-52), DNMs on sequences the file does
    not have ask for nothing; no index, UZ_IO_INDEX=0 or the Python decoders: None (decode the file)."""
    from filesio import write_bgzf_text, write_tbi
    from unfazed_amd import session
    path = os.path.join(str(tmp_path), "sites.vcf.gz")
    write_bgzf_text(path, _big_vcf_text(n_per_contig=300), block_bytes=20000)
    dnms = [dict(chrom="1", start=50_000, end=50_001), dict(chrom="chrX", start=1200, end=90_000), dict(chrom="7", start=5, end=6),
            dict(chrom="chr2", start=100, end=101)]
    assert session.site_regions(path, dnms, 5000) is None  # no index yet
    write_tbi(path)
    got = session.site_regions(path, dnms, 5000)
    assert got == ((0, 44_998, 55_003), (1, 0, 5_103), (2, 0, 95_002))
    monkeypatch.setenv("UZ_IO_INDEX", "0")
    assert session.site_regions(path, dnms, 5000) is None
    monkeypatch.delenv("UZ_IO_INDEX")
    key, table = session.load_sites(path, got)
    whole = io_native.read_vcf_table(path)
    assert "@" in key and 0 < table.pos.size < whole.pos.size and table.contigs == whole.contigs
    del session._SITES[key]
