"""uz_bam_stage_* (BAM file -> staged records in one pass) against the three-step path it replaces -- uz_bam_decode_regions
(ASCII table) -> uz_reads_pack -> uz_reads_select_* -- on the same fetches: every column byte for byte."""
import numpy as np
import pytest

from synth import bigsynth
from synth.sites_np import make_clusters, make_sites, place_dnms_full
from unfazed_amd import abi, io_native


def three_step(bam, fc, flo, fhi, fex, mbq, all_bases=False, lists=True, wide_no_units=False):
    table = io_native.read_bam_regions(bam, fc, flo, fhi, threads=2, insert_size_max_sample=0)
    v = abi.reads_view(table)
    full = io_native.pack_reads(v, mbq, lists=True, with_end=True)
    return io_native.ReadsSource(full).select(fc, flo, fhi, all_bases=all_bases, lists=lists, extra=fex, wide_no_units=wide_no_units), table


def assert_same(a, b):
    va, vb = a.view, b.view
    for f, _ in abi.ReadsPackedView._fields_:
        x, y = getattr(va, f), getattr(vb, f)
        if isinstance(x, int) and not f.startswith(("contig_off", "max_span")) and f not in a.arrays and f not in b.arrays:
            assert x == y, f
    assert set(a.arrays) == set(b.arrays), (sorted(a.arrays), sorted(b.arrays))
    n = int(va.n_segs)
    sizes = {"cigar": int(va.n_cigar_total), "seq2": 8 * int(va.n_seq_units), "exc_rec": int(va.n_exc), "exc_pos": int(va.n_exc), "exc_code": int(va.n_exc),
             "qlow_pos": int(va.n_qlow_pos) * (2 if va.qlow_pos_wide else 1), "esc16_key": int(va.n_esc16), "esc16_val": int(va.n_esc16),
             "qlow": 4 * int(va.n_row_units), "contig_off": int(va.n_contigs) + 1, "max_span": int(va.n_contigs),
             "tup_hot": 256, "tup_esc": int(va.n_tup_esc), "tup_esc_off": (n + abi.TUP8_SPAN - 1) // abi.TUP8_SPAN + 1}
    for k in a.arrays:
        m = sizes.get(k, int(va.n_tup) if k.startswith("tup_") else n)
        assert np.array_equal(a.arrays[k][:m], b.arrays[k][:m]), k


def fetches_of(w, stride=1, spread=5):
    """fetch points as staging.fetch_points lists them: the DNM ([start - 1, start + 1), extra = allele length) and a few
    one-base fetches at 'het sites' of its window"""
    dn = w["dn"]
    rng = np.random.default_rng(5)
    c, lo, hi, ex = [], [], [], []
    for i in range(0, dn.n, stride):
        c.append(dn.contig[i]); lo.append(dn.start[i] - 1); hi.append(dn.start[i] + 1); ex.append(max(len(dn.refs[i]), len(dn.alts[i])))
        for p in np.sort(rng.integers(dn.start[i] - 5000, dn.start[i] + 5000, spread)):
            c.append(dn.contig[i]); lo.append(int(p)); hi.append(int(p) + 1); ex.append(0)
    return (np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32), np.array(ex, np.uint16))


@pytest.mark.parametrize("stride,spread", [(1, 5), (3, 9), (7, 0)])
def test_unit_masked_lists_equal_three_step(workload, stride, spread):
    fc, flo, fhi, fex = fetches_of(workload, stride, spread)
    want, table = three_step(workload["bam"], fc, flo, fhi, fex, 20)
    src = io_native.BamSource(workload["bam"], threads=3)
    got = src.select(fc, flo, fhi, 20, extra=fex)
    assert_same(got, want)
    # names by id
    n = int(got.view.n_segs)
    q = abi.wide_columns(got)["qname"]
    for i in range(0, n, max(1, n // 200)):
        assert got.qnames[int(q[i])] == table.qnames[int(table.qname[i])]
    assert got.io_stats["records_kept"] == n
    assert np.array_equal(src.tlen_head[:100], io_native.read_bam_table(workload["bam"], threads=2).tlen_head[:100])


def test_all_bases_and_other_thresholds(workload):
    fc, flo, fhi, fex = fetches_of(workload, 2, 4)
    src = io_native.BamSource(workload["bam"], threads=2)
    for mbq in (13, 38):
        want, _ = three_step(workload["bam"], fc, flo, fhi, fex, mbq)
        assert_same(src.select(fc, flo, fhi, mbq, extra=fex), want)
    want, _ = three_step(workload["bam"], fc, flo, fhi, fex, 20, all_bases=True)
    assert_same(src.select(fc, flo, fhi, 20, extra=fex, all_bases=True), want)
    want, _ = three_step(workload["bam"], fc, flo, fhi, None, 20)
    assert_same(src.select(fc, flo, fhi, 20), want)


def test_wide_fetches_keep_every_unit(workload):
    """SV-style fetches (+-cutoff around a breakpoint): no unit masks for their records"""
    dn = workload["dn"]
    fc = dn.contig[::4].astype(np.int32)
    flo = (dn.start[::4] - 700).astype(np.int32)
    fhi = (dn.start[::4] + 700).astype(np.int32)
    fex = np.zeros(fc.size, np.uint16)
    want, _ = three_step(workload["bam"], fc, flo, fhi, fex, 20)
    got = io_native.BamSource(workload["bam"], threads=2).select(fc, flo, fhi, 20, extra=fex)
    assert_same(got, want)


def test_empty_and_absent(workload):
    src = io_native.BamSource(workload["bam"], threads=2)
    got = src.select(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32), 20)
    assert int(got.view.n_segs) == 0
    # a fetch where the file has no records, a contig id out of range
    got = src.select(np.array([2, 7], np.int32), np.array([900_000, 5], np.int32), np.array([900_001, 6], np.int32), 20)
    assert int(got.view.n_segs) == 0


@pytest.mark.parametrize("slack", ["0", "40", "300"])
def test_mates_outside_the_reach_go_through_the_index(workload, slack, monkeypatch):
    """with a small slack around the fetch points most mates lie outside every reach interval: they are answered by another
    task or looked up through the index -- the table must not change"""
    fc, flo, fhi, fex = fetches_of(workload, 4, 3)  # (every look-up through the index is a walk of its own: a batch the CPU suite can afford)
    want, _ = three_step(workload["bam"], fc, flo, fhi, fex, 20)
    monkeypatch.setenv("UZ_STAGE_SLACK", slack)
    got = io_native.BamSource(workload["bam"], threads=3).select(fc, flo, fhi, 20, extra=fex)
    assert got.io_stats["index_mate_lookups"] > 0
    assert_same(got, want)


@pytest.mark.parametrize("seed,readlen", [(31, 151), (32, 100), (33, 301)])
def test_odd_records_from_a_python_written_bam(tmp_path, seed, readlen):
    """The other writer, the other kind of data: synth.small's pile-ups (duplicates, QC failures, secondary and supplementary
    records with SA tags, mates on other contigs or unmapped, overlapping mates, many-operation CIGARs, low-quality runs, reads
    of 100 and 301 bases) written by the Python BAM writer + write_bai, staged in one pass and through the three steps."""
    from filesio import dump_dataset, write_bai
    from synth.small import SmallConfig, make_small
    ds = make_small(SmallConfig(seed=seed, n_dnms=10, odd_read_prob=0.25, lowq_prob=0.08, softclip_prob=0.05, indel_prob=0.03, readlen=readlen, cluster_prob=0.6))
    paths = dump_dataset(ds, str(tmp_path))
    bam = list(paths["bams"].values())[0]
    write_bai(bam)
    full = io_native.read_bam_table(bam, threads=2)
    assert (np.asarray(full.flag) & 0x900).any() and (np.asarray(full.flag) & 0x8).any()  # secondary / supplementary, mate unmapped
    rng = np.random.default_rng(seed)
    c, lo, hi, ex = [], [], [], []
    for d in ds.dnms:
        tid = full.contig_index[d["chrom"]]
        c.append(tid); lo.append(d["start"] - 1); hi.append(d["start"] + 1); ex.append(max(1, d["end"] - d["start"]))
        for p in np.sort(rng.integers(d["start"] - 5000, d["start"] + 5000, 8)):
            c.append(tid); lo.append(int(p)); hi.append(int(p) + 1); ex.append(0)
    fc, flo, fhi, fex = np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32), np.array(ex, np.uint16)
    src = io_native.BamSource(bam, threads=3)
    for kw in (dict(extra=fex), dict(extra=fex, all_bases=True), dict()):
        want, _ = three_step(bam, fc, flo, fhi, kw.get("extra"), 20, all_bases=kw.get("all_bases", False))
        assert_same(src.select(fc, flo, fhi, 20, **kw), want)
    # wide (SV-style) fetches over the same file
    wc, wlo, whi = fc[::9], (flo[::9] - 600).astype(np.int32), (fhi[::9] + 600).astype(np.int32)
    want, _ = three_step(bam, wc, wlo, whi, np.zeros(wc.size, np.uint16), 20)
    assert_same(src.select(wc, wlo, whi, 20, extra=np.zeros(wc.size, np.uint16)), want)


def test_reads_longer_than_the_reach_slack(tmp_path):
    """2.5 kb reads: a record can overlap the fetch points of two reach intervals and a mate lies beyond any slack -- the kept
    records of neighbouring tasks interleave in the file (sorted and folded by virtual offset), mates go through the index"""
    from filesio import dump_dataset, write_bai
    from synth.small import SmallConfig, make_small
    ds = make_small(SmallConfig(seed=41, n_dnms=6, odd_read_prob=0.1, readlen=2500, coverage_per_hap=4.0, ins_mean=7500, ins_sd=100))
    paths = dump_dataset(ds, str(tmp_path))
    bam = list(paths["bams"].values())[0]
    write_bai(bam)
    full = io_native.read_bam_table(bam, threads=2)
    rng = np.random.default_rng(5)
    c, lo, hi, ex = [], [], [], []
    for d in ds.dnms:
        tid = full.contig_index[d["chrom"]]
        c.append(tid); lo.append(d["start"] - 1); hi.append(d["start"] + 1); ex.append(1)
        for p in np.sort(rng.integers(d["start"] - 5000, d["start"] + 5000, 3)):
            c.append(tid); lo.append(int(p)); hi.append(int(p) + 1); ex.append(0)
    fc, flo, fhi, fex = np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32), np.array(ex, np.uint16)
    want, _ = three_step(bam, fc, flo, fhi, fex, 20)
    got = io_native.BamSource(bam, threads=3).select(fc, flo, fhi, 20, extra=fex)
    assert got.io_stats["index_mate_lookups"] > 0
    assert_same(got, want)


def test_sv_batch_form_wide_fetches_stage_no_units(workload):
    """an SV batch: +-cutoff fetches around two breakpoints (no base unit for what only they return) mixed with the one-base fetches at
    het sites of the windows (their units, also on records a wide fetch returns too)"""
    dn = workload["dn"]
    rng = np.random.default_rng(8)
    c, lo, hi, ex = [], [], [], []
    for i in range(0, dn.n, 3):
        for bp in (int(dn.start[i]), int(dn.start[i]) + 3000):
            c.append(dn.contig[i]); lo.append(max(bp - 500, 0)); hi.append(bp + 500); ex.append(0)
        for p in np.sort(rng.integers(dn.start[i] - 5000, dn.start[i] + 5000, 6)):
            c.append(dn.contig[i]); lo.append(int(p)); hi.append(int(p) + 1); ex.append(0)
    fc, flo, fhi, fex = np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32), np.array(ex, np.uint16)
    want, _ = three_step(workload["bam"], fc, flo, fhi, fex, 20, wide_no_units=True)
    got = io_native.BamSource(workload["bam"], threads=3).select(fc, flo, fhi, 20, extra=fex, wide_no_units=True)
    assert_same(got, want)
    plain, _ = three_step(workload["bam"], fc, flo, fhi, fex, 20)
    assert int(got.view.n_seq_units) < int(plain.view.n_seq_units) // 2  # most of the records only a wide fetch returns keep no unit
    um = got.arrays["tup_umask"][abi.tup_column(got)]
    assert (um == 0).sum() > int(got.view.n_segs) // 3


def test_blocks_inflated_elsewhere_give_the_same_batch(workload, monkeypatch):
    """uz_bam_stage_begin / uz_stage_gather_blocks / uz_stage_set_inflated / uz_bam_stage_finish: the blocks the walk will read, gathered
    and inflated by somebody else (here zlib, block by block; in the product the device, tests/test_inflate_gpu.py), give the same batch
    byte for byte; blocks the gather did not list (mates through the index) still go through the host's inflate; a block that comes
    back wrong is refused by its CRC-32."""
    import zlib
    w = workload
    fc, flo, fhi, fex = fetches_of(w, 2, 4)
    src = io_native.BamSource(w["bam"], threads=2)
    plain = src.select(fc, flo, fhi, 20, extra=fex)
    calls = []

    def host_inflate(comp, comp_bytes, in_off, out_off, out, spoil=False):
        calls.append((int(in_off.size), int(comp_bytes), int(out_off[-1])))
        for k in range(in_off.size):
            n = int(out_off[k + 1] - out_off[k])
            data = zlib.decompressobj(-15).decompress(bytes(comp[int(in_off[k]): min(int(comp_bytes), int(in_off[k]) + 70000)]), n) if n else b""  # (a BGZF block is at most 64 KiB)
            assert len(data) == n
            out[int(out_off[k]): int(out_off[k]) + n] = np.frombuffer(data, np.uint8)
        if spoil:
            out[int(out_off[-1]) // 2] ^= 1

    got = src.select(fc, flo, fhi, 20, extra=fex, inflate=host_inflate)
    assert_same(got, plain)
    assert calls and calls[0][0] > 10 and got.io_stats["blocks_from_the_device"] >= 0.9 * got.io_stats["blocks_inflated"] > 0
    assert got.pre_inflate["blocks"] == calls[0][0] and got.pre_inflate["out_bytes"] == calls[0][2]
    # mates outside the reach: looked up through the index, in blocks the gather does not list
    monkeypatch.setenv("UZ_STAGE_SLACK", "50")
    far = io_native.BamSource(w["bam"], threads=2)
    fc, flo, fhi, fex = fetches_of(w, 5, 2)  # (every look-up through the index is a walk of its own: a smaller batch)
    a = far.select(fc, flo, fhi, 20, extra=fex)
    b = far.select(fc, flo, fhi, 20, extra=fex, inflate=host_inflate)
    assert_same(a, b)
    assert b.io_stats["index_mate_lookups"] > 0 and b.io_stats["blocks_from_the_device"] < b.io_stats["blocks_inflated"]
    monkeypatch.delenv("UZ_STAGE_SLACK")
    with pytest.raises(io_native.IoError, match="CRC mismatch in the pre-inflated"):
        src.select(fc, flo, fhi, 20, extra=fex, inflate=lambda *x: host_inflate(*x, spoil=True))
