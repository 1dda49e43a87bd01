"""The record walk on the device (include/uz_bamwalk.h, csrc/k_bamwalk.hip) through the C ABI:
  * k_bam_walk's descriptors == the host's twin (uz_stage_walk_host) on the same plan, field by field -- the host's walk (io_stage.cpp: walk_task) is
    what `bamfile.fetch(chrom, lo, hi)` iterates for the reference (read_collector.py:385, :167);
  * the table uz_reads_from_bam builds from the bytes in HBM == the table the host route stages for the same fetches: every header the device holds
    (uz_reads_headers), then the read stage's answers on both tables.
Sizes the host finishes in seconds; the full-size feed is bench.py's `feed` leg."""
import numpy as np
import pytest

from unfazed_amd import abi, io_native
from test_io_stage import fetches_of

pytestmark = pytest.mark.gpu

FIELDS = ("voff", "src", "h1", "pos", "end", "tlen", "mpos", "mtid", "h2", "task", "flag", "l_seq", "n_cigar", "mapq", "l_name", "direct")


def walk_both(engine, bam, fc, flo, fhi, fex, all_bases=False):
    src = io_native.BamSource(bam, threads=3)
    twin = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, small_tasks=True)
    dev = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, walk=engine.bam_walk, merge=True)  # (merge: the descriptors per task of the stage, as the twin's)
    return src, twin, dev


@pytest.mark.parametrize("stride,spread", [(1, 5), (3, 9), (7, 0)])
def test_descriptors_equal_the_hosts_walk(engine, workload, stride, spread):
    fc, flo, fhi, fex = fetches_of(workload, stride, spread)
    src, twin, dev = walk_both(engine, workload["bam"], fc, flo, fhi, fex)
    try:
        flagged = np.nonzero(dev.d_flags)[0]
        assert (dev.d_flags & 2).sum() == 0  # no malformed record in a well-formed file
        ok = np.ones(twin.desc.size, bool)
        for t in flagged:  # a task handed back to the host owns no descriptors on the device
            ok[twin.d_first[t]: twin.d_first[t + 1]] = False
        # the device hands over only what the host's joins can need: the direct records and those that share a name hash with a direct record
        # of their task (io_stage.cpp finish_task: "mate candidates"; the device's set merges hashes that differ in bit 0 only)
        for t in range(twin.d_first.size - 1):
            d = twin.desc[twin.d_first[t]: twin.d_first[t + 1]]
            keys = np.unique(d["h1"][d["direct"] != 0] | np.uint64(1))
            ok[twin.d_first[t]: twin.d_first[t + 1]] &= (d["direct"] != 0) | np.isin(d["h1"] | np.uint64(1), keys)
        want = twin.desc[ok]
        assert want.size < twin.desc.size
        assert dev.desc.size == want.size
        for f in FIELDS:
            assert np.array_equal(dev.desc[f], want[f]), f
        assert flagged.size <= max(2, twin.d_first.size // 10), "too many tasks fall back to the host: %d of %d" % (flagged.size, twin.d_first.size - 1)
        # the list of kept records is the same whoever walked
        for f in ("qname", "mate", "cig_off", "unit_off", "seq_off"):
            assert np.array_equal(dev.kept[f], twin.kept[f]), f
        same_place = (dev.kept["src"] & np.uint64(io_native.WALK_SRC_AUX)) == 0
        assert np.array_equal(dev.kept["src"][same_place], twin.kept["src"][same_place])
    finally:
        engine.bam_walk_release(dev.token)
        dev.token = None


@pytest.mark.parametrize("all_bases", [False, True])
def test_table_from_hbm_equals_the_staged_table(engine, workload, all_bases):
    fc, flo, fhi, fex = fetches_of(workload, 1, 5)
    src = io_native.BamSource(workload["bam"], threads=3)
    staged = src.select(fc, flo, fhi, 20, extra=fex, all_bases=all_bases)
    rid_a = engine.upload_reads_packed(staged)
    engine.wait_reads(rid_a)
    dev = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, walk=engine.bam_walk)
    rid_b = engine.reads_from_bam(dev, names=True)
    try:
        n = int(staged.view.n_segs)
        assert dev.n == n
        a, b = engine.reads_headers(rid_a, n), engine.reads_headers(rid_b, n)
        for k in ("start", "end", "tlen", "mate", "qname"):
            assert np.array_equal(a[k], b[k]), k
        # the read names, fetched from HBM: id by id those of the host route
        ids = np.arange(len(staged.qnames), dtype=np.uint32)
        assert len(dev.qnames) == ids.size
        assert dev.qnames.take(ids) == staged.qnames.take(ids)
        assert dev.qnames[3] == staged.qnames[3]
    finally:
        engine.free_reads(rid_a)
        engine.free_reads(rid_b)


def test_mates_through_the_index_and_flagged_tasks(engine, workload, monkeypatch):
    """a small slack: mates outside every reach interval are looked up through the index by the HOST -- their bytes travel as aux bytes"""
    fc, flo, fhi, fex = fetches_of(workload, 2, 3)
    monkeypatch.setenv("UZ_STAGE_SLACK", "40")
    src = io_native.BamSource(workload["bam"], threads=3)
    staged = src.select(fc, flo, fhi, 20, extra=fex)
    rid_a = engine.upload_reads_packed(staged)
    engine.wait_reads(rid_a)
    dev = src.select_kept(fc, flo, fhi, 20, walk=engine.bam_walk)
    assert dev.n_aux > 0 and dev.io_stats["index_mate_lookups"] > 0
    rid_b = engine.reads_from_bam(dev)
    try:
        n = int(staged.view.n_segs)
        a, b = engine.reads_headers(rid_a, n), engine.reads_headers(rid_b, n)
        for k in ("start", "end", "tlen", "mate", "qname"):
            assert np.array_equal(a[k], b[k]), k
    finally:
        engine.free_reads(rid_a)
        engine.free_reads(rid_b)


def test_a_block_that_fails_its_crc_is_refused(engine, workload):
    """the device holds every inflated block against the CRC-32 of its BGZF footer (k_bgzf_crc32), as htslib's reader does"""
    from unfazed_amd.engine import UnfazedHipError
    fc, flo, fhi, fex = fetches_of(workload, 3, 3)
    src = io_native.BamSource(workload["bam"], threads=3)

    def walk(plan):
        plan["blk_crc"] = plan["blk_crc"].copy()
        plan["blk_crc"][5] ^= np.uint32(0x80)
        return engine.bam_walk(plan)

    with pytest.raises(UnfazedHipError, match="CRC mismatch in BGZF block 5"):
        src.select_kept(fc, flo, fhi, 20, walk=walk)
    # ... and the batch after it goes through (the failed walk gave its slot back)
    dev = src.select_kept(fc, flo, fhi, 20, walk=engine.bam_walk, release=engine.bam_walk_release)
    assert dev.n > 0
    del dev


def test_a_kept_list_that_points_beyond_its_stores_is_refused(engine, workload):
    """the kept list is the caller's: k_bam_extract writes no record beyond the totals it came with"""
    from unfazed_amd.engine import UnfazedHipError
    fc, flo, fhi, fex = fetches_of(workload, 3, 3)
    src = io_native.BamSource(workload["bam"], threads=3)
    for col, val in (("cig_off", 2 ** 31), ("seq_off", 2 ** 31), ("unit_off", 2 ** 31), ("src", np.uint64(2 ** 40))):
        dev = src.select_kept(fc, flo, fhi, 20, walk=engine.bam_walk, release=engine.bam_walk_release)
        k = int(np.nonzero(dev.kept["seq_off"] != io_native.KEPT_NO_SEQ)[0][-1])
        dev.kept[col][k] = val
        with pytest.raises(UnfazedHipError, match="kept"):
            engine.reads_from_bam(dev)
        del dev  # (gives the walked batch up)
    dev = src.select_kept(fc, flo, fhi, 20, walk=engine.bam_walk, release=engine.bam_walk_release)
    rid = engine.reads_from_bam(dev)
    engine.free_reads(rid)
