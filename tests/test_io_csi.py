"""Region decode through a CSI index: a BCF next to NAME.bcf.csi (what `bcftools index` writes; the reference hands a .bcf to
cyvcf2 and asks it per DNM region, informative_site_finder.py:41-43, :213) and a BGZF VCF next to NAME.vcf.gz.csi (`tabix -C`), for the
usual 14 / 5 scheme and for a deeper one -- held against the whole-file decode restricted to the same records, and against the TBI route."""
import os

import numpy as np
import pytest

from filesio import write_bgzf_text, write_csi, write_tbi
from test_io_native import _big_vcf_text
from unfazed_amd import io_native, session


def _records_of(text):
    """the records of a VCF text as the objects tests/bcfio.write_bcf takes"""
    class R:
        pass
    samples, contigs, recs = None, [], []
    for line in text.split("\n"):
        if line.startswith("##contig=<ID="):
            contigs.append(line[13:].split(",")[0].rstrip(">"))
        elif line.startswith("#CHROM"):
            samples = line.split("\t")[9:]
        elif line and not line.startswith("#"):
            f = line.split("\t")
            r = R()
            r.chrom, r.start, r.ref, r.alts = f[0], int(f[1]) - 1, f[3], ([] if f[4] == "." else f[4].split(","))
            r.info = {} if f[7] == "." else dict(kv.split("=") for kv in f[7].split(";"))
            r.end = int(r.info["END"]) if "END" in r.info else None
            gts, rds, ads, gqs = [], [], [], []
            for col in f[9:]:
                gt, ad, gq = col.split(":")
                a = gt.replace("|", "/").split("/")
                gts.append(2 if "." in a else {("0", "0"): 0, ("0", "1"): 1, ("1", "0"): 1, ("1", "1"): 3}[tuple(a)])
                rd_, ad_ = (ad.split(",") + ["."])[:2] if ad != "." else (".", ".")
                rds.append(-1 if rd_ == "." else int(rd_))
                ads.append(-1 if ad_ == "." else int(ad_))
                gqs.append(-1.0 if gq == "." else float(gq))
            r.gt_types, r.ref_depths, r.alt_depths, r.gt_quals = gts, rds, ads, gqs
            recs.append(r)
    for c in [r.chrom for r in recs]:
        if c not in contigs:
            contigs.append(c)
    return samples, contigs, recs


def _held_against_the_whole_file(path, whole, names, seed, lines=True):
    rng = np.random.default_rng(seed)
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
    for k in ("pos", "end", "sflags", "ref_base", "alt_base"):
        assert np.array_equal(getattr(whole, k)[idx], getattr(t, k)), k
    for k in ("gt", "ref_depth", "alt_depth", "gq"):
        assert np.array_equal(getattr(whole, k)[:, idx], getattr(t, k)), k
    assert t.samples == whole.samples
    assert [c for c in whole.contigs if c in t.contigs] == t.contigs
    for j in range(0, idx.size, 37):
        i = int(idx[j])
        assert whole.ref_str[i] == t.ref_str[j] and whole.alt_strs[i] == t.alt_strs[j]
        if lines:
            assert whole.lines[i] == t.lines[j]
    assert t.io_stats[0] < 0.4 * os.path.getsize(path) and t.io_stats[3] == idx.size and t.io_stats[2] < 0.4 * whole.pos.size
    return t, idx


@pytest.mark.parametrize("min_shift,depth", [(14, 5), (12, 6), (16, 3)])
def test_vcf_region_decode_through_a_csi_index(tmp_path, min_shift, depth):
    path = os.path.join(str(tmp_path), "sites.vcf.gz")
    write_bgzf_text(path, _big_vcf_text(), block_bytes=20000)
    write_csi(path, min_shift=min_shift, depth=depth)
    assert io_native.tabix_index_path(path) == path + ".csi"
    names = io_native.tabix_contigs(path)
    assert names == ["chr1", "chr2", "chrX"]
    whole = io_native.read_vcf_table(path)
    t, _ = _held_against_the_whole_file(path, whole, names, 8)
    assert (whole.end - whole.pos).max() > 1000  # (long records reach into windows: the higher bins are asked too)
    # ... and the same table as through the TBI of the same file
    write_tbi(path)
    assert io_native.tabix_index_path(path) == path + ".tbi"
    t2, _ = _held_against_the_whole_file(path, whole, names, 8)
    assert np.array_equal(t.pos, t2.pos) and np.array_equal(t.gt, t2.gt)
    empty = io_native.read_vcf_table_regions(path, [], [], [], tbi=path + ".csi")
    assert empty.pos.size == 0 and empty.samples == whole.samples
    far = io_native.read_vcf_table_regions(path, [2], [2_000_000_000 - 10], [2_000_000_000], tbi=path + ".csi")
    assert far.pos.size == 0
    with pytest.raises(io_native.IoError):
        io_native.read_vcf_table_regions(path, [7], [0], [10], tbi=path + ".csi")


@pytest.mark.parametrize("use_idx", [False, True])
def test_bcf_region_decode_through_its_csi_index(tmp_path, use_idx):
    from bcfio import write_bcf
    samples, contigs, recs = _records_of(_big_vcf_text())
    contigs = ["chrUn_first"] + contigs  # (a header contig without records: the index's references are the header's ids, not the order of appearance)
    path = os.path.join(str(tmp_path), "sites.bcf")
    write_bcf(path, samples, recs, contigs, use_idx=use_idx)
    assert io_native.tabix_index_path(path) is None
    write_csi(path)
    assert io_native.tabix_index_path(path) == path + ".csi"
    names = io_native.tabix_contigs(path)
    assert names == contigs
    whole = io_native.read_vcf_table(path)
    assert whole.is_bcf and whole.contigs == ["chr1", "chr2", "chrX"]
    t, idx = _held_against_the_whole_file(path, whole, names, 9, lines=False)
    assert t.is_bcf
    svt = [t.info(j, "SVTYPE") for j in range(t.pos.size)]
    assert svt == [whole.info(int(i), "SVTYPE") for i in idx] and any(x is not None for x in svt)
    empty = io_native.read_vcf_table_regions(path, [], [], [])
    assert empty.pos.size == 0 and empty.samples == whole.samples and empty.is_bcf
    nothing = io_native.read_vcf_table_regions(path, [0], [0], [1_000_000])  # the contig without records
    assert nothing.pos.size == 0
    with pytest.raises(io_native.IoError):
        io_native.read_vcf_table_regions(path, [len(contigs) + 3], [0], [10])


def test_session_takes_a_bcf_through_its_csi_index(tmp_path, monkeypatch):
    from bcfio import write_bcf
    samples, contigs, recs = _records_of(_big_vcf_text())
    path = os.path.join(str(tmp_path), "sites.bcf")
    write_bcf(path, samples, recs, contigs)
    dnms = [dict(chrom="1", start=int(recs[50].start), end=int(recs[50].start) + 1), dict(chrom="chr2", start=120_000, end=120_001)]
    monkeypatch.setattr(session, "_SITES", {})
    assert session.site_regions(path, dnms, 5000) is None  # no index: the file is decoded whole
    write_csi(path)
    got = session.site_regions(path, dnms, 5000)
    assert got is not None and len(got) == 2 and {g[0] for g in got} == {contigs.index("chr1"), contigs.index("chr2")}
    key, table = session.load_sites(path, got)
    whole = io_native.read_vcf_table(path)
    assert 0 < table.pos.size < 0.05 * whole.pos.size


def test_corrupt_csi_indexes_are_refused(tmp_path):
    import struct
    import zlib
    from unfazed_amd.io_bam import _bgzf_block
    path = os.path.join(str(tmp_path), "sites.vcf.gz")
    write_bgzf_text(path, _big_vcf_text(), block_bytes=20000)
    csi = write_csi(path)
    from filesio import _inflated_blocks
    _, raw = _inflated_blocks(csi)

    def put(blob):
        with open(csi, "wb") as fh:
            for i in range(0, len(blob), 60000):
                fh.write(_bgzf_block(blob[i: i + 60000]))
            fh.write(_bgzf_block(b""))

    l_aux = struct.unpack_from("<i", raw, 12)[0]
    at_nref = 16 + l_aux
    cases = {
        "truncated": raw[: at_nref + 4 + 4 + 10],
        "bad scheme": raw[:4] + struct.pack("<ii", 40, 9) + raw[12:],
        "negative bins": raw[: at_nref + 4] + struct.pack("<i", -5) + raw[at_nref + 8:],
        "huge chunk count": raw[: at_nref + 8 + 12] + struct.pack("<i", 1 << 30) + raw[at_nref + 8 + 16:],
        "aux beyond the file": raw[:12] + struct.pack("<i", 1 << 28) + raw[16:],
        "not an index": b"XYZ\1" + raw[4:],
    }
    for what, blob in cases.items():
        put(bytes(blob))
        with pytest.raises(io_native.IoError):
            io_native.read_vcf_table_regions(path, [0], [0], [100000])
        with pytest.raises(io_native.IoError):
            io_native.tabix_contigs(path)
    put(raw)
    assert io_native.read_vcf_table_regions(path, [0], [0], [100000]).pos.size > 0


@pytest.mark.parametrize("name", ["trio_hets_snvs_chr22.vcf.gz", "trio_hets_svs_chr22.vcf.gz", "trio_svs_chr22.vcf.gz"])
@pytest.mark.parametrize("min_shift,depth", [(14, 5), (13, 6)])
def test_csi_route_equals_the_real_tbi_route_on_the_reference_files(tmp_path, name, min_shift, depth):
    """files bgzip wrote and tabix indexed (the reference's test/data, held as fixtures under tests/golden/refdata): a CSI of this repo's writer
    next to a copy of the file gives, interval set by interval set, the table the file's own TBI gives"""
    import shutil
    here = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(here, "golden", "refdata", name)
    path = os.path.join(str(tmp_path), name)
    shutil.copy(src, path)
    write_csi(path, min_shift=min_shift, depth=depth)
    assert io_native.tabix_index_path(path) == path + ".csi" and io_native.tabix_index_path(src) == src + ".tbi"
    assert io_native.tabix_contigs(path) == io_native.tabix_contigs(src)
    full = io_native.read_vcf_table(src, threads=2)
    pos, end = np.asarray(full.pos, np.int64), np.asarray(full.end, np.int64)
    rng = np.random.default_rng(len(name) + depth)
    lo_all, hi_all = int(pos.min()), int(end.max())
    kept = 0
    for trial in range(120):
        k = int(rng.integers(1, 6))
        width = int(rng.choice([1, 10, 1000, 20000, 300000]))
        lo = rng.integers(max(0, lo_all - 5000), hi_all + 5000, k)
        hi = lo + rng.integers(1, width + 1, k)
        ref = np.zeros(k, np.int32)
        a = io_native.read_vcf_table_regions(src, ref, lo, hi, threads=2)
        b = io_native.read_vcf_table_regions(path, ref, lo, hi, threads=2)
        for col in ("pos", "end", "gt", "ref_depth", "alt_depth", "gq", "sflags"):
            assert np.array_equal(getattr(a, col), getattr(b, col)), (trial, col, lo, hi)
        assert list(a.lines) == list(b.lines)
        kept += int(a.pos.size)
    assert kept > 0  # (the snvs file holds a few dozen records: what counts is that every set agrees)
