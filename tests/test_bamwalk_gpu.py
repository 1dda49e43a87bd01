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


def _long_insert_bam(path):
    """pairs 3 kb apart (template length far beyond the 1 kb reach slack) tiling 10 kb - 62 kb of one contig: the mate of a record a fetch returns
    lies in ANOTHER reach interval of the same stage task -- and, where the walk plan cuts that task into sub-tasks of at most 32 kb, in
    another sub-task"""
    from filesio import write_bai
    from unfazed_amd.io_bam import write_bam
    from unfazed_amd.model import Segment
    segs = []
    L = 151
    for k, p in enumerate(range(10_000, 59_000, 23)):
        q = "pair%05d" % k
        m = p + 3000
        segs.append(Segment(q, 0x1 | 0x2 | 0x20 | 0x40, 0, p, 60, [(0, L)], 0, m, m + L - p, "ACGT" * 37 + "ACG", [37] * L))
        segs.append(Segment(q, 0x1 | 0x2 | 0x10 | 0x80, 0, m, 60, [(0, L)], 0, p, -(m + L - p), "TGCA" * 37 + "TGC", [37] * L))
    segs.sort(key=lambda s: s.pos)
    write_bam(path, [("1", 200_000)], segs)
    write_bai(path)


def test_a_mate_in_another_sub_task_of_the_same_stage_task_is_kept(engine, tmp_path):
    """the device built its mate-candidate hash set per walk task -- a sub-task of the stage's task -- where the host's rule
    (finish_task) works per stage task: a non-direct mate walked by sub-task u whose direct partner lay in sub-task u + 1 was dropped on the
    device route and never looked up through the index.  The set now belongs to the stage task: the device's descriptors are again exactly the
    host rule's, and every direct record keeps its mate."""
    bam = str(tmp_path / "long.bam")
    _long_insert_bam(bam)
    pts = np.arange(12_000, 58_001, 2_500)  # one-base fetches 2.5 kb apart: reach intervals that do not merge, one stage task, two sub-tasks
    fc = np.zeros(pts.size, np.int32)
    flo, fhi, fex = pts.astype(np.int32), (pts + 1).astype(np.int32), np.zeros(pts.size, np.uint16)
    src = io_native.BamSource(bam, threads=2)
    twin = src.select_kept(fc, flo, fhi, 20, small_tasks=True)
    dev = src.select_kept(fc, flo, fhi, 20, walk=engine.bam_walk, merge=True)
    try:
        assert dev.plan["task"].reshape(-1, 10).shape[0] > np.unique(dev.plan["task"].reshape(-1, 10)[:, 9]).size, "the plan did not cut a stage task into sub-tasks"
        assert (dev.d_flags != 0).sum() == 0
        ok = np.ones(twin.desc.size, bool)
        for t in range(twin.d_first.size - 1):
            d = twin.desc[twin.d_first[t]: twin.d_first[t + 1]]
            keys = np.unique(d["h1"][d["direct"] != 0] | np.uint64(1))
            ok[twin.d_first[t]: twin.d_first[t + 1]] &= (d["direct"] != 0) | np.isin(d["h1"] | np.uint64(1), keys)
        want = twin.desc[ok]
        assert dev.desc.size == want.size
        for f in FIELDS:
            assert np.array_equal(dev.desc[f], want[f]), f
        for f in ("qname", "mate", "cig_off", "unit_off", "seq_off"):
            assert np.array_equal(dev.kept[f], twin.kept[f]), f
        # every record a fetch returns has its mate in the kept list (the mates lie 3 kb away: inside the stage task's reach, outside the record's own interval)
        assert (dev.kept["mate"] >= 0).all()
    finally:
        engine.bam_walk_release(dev.token)
        dev.token = None


def test_a_batch_beyond_the_walk_cap_takes_the_host_route(engine, workload, monkeypatch):
    """the device walk keeps a batch's inflated bytes and worst-case descriptor slices in HBM with no size cap.  A batch whose blocks
    inflate to more than UZ_WALK_MAX_BYTES is staged by the host (the link form) instead of failing in hipMalloc: same table either way."""
    fc, flo, fhi, fex = fetches_of(workload, 2, 4)
    src = io_native.BamSource(workload["bam"], threads=3)
    rid_a, names_a = engine.upload_reads_staged(src, fc, flo, fhi, fex, 20)
    monkeypatch.setenv("UZ_WALK_MAX_BYTES", "1")
    rid_b, names_b = engine.upload_reads_staged(src, fc, flo, fhi, fex, 20)
    monkeypatch.delenv("UZ_WALK_MAX_BYTES")
    try:
        assert "joins" in names_a.timing and "joins" not in names_b.timing  # (the device walk's timing keys; the host stage's: spans / walk / mates / ...)
        n = int(names_b.io_stats["records_kept"])
        assert n == int(names_a.io_stats["records_kept"])
        a, b = engine.reads_headers(rid_a, n), engine.reads_headers(rid_b, n)
        for k in ("start", "end", "tlen", "mate", "qname"):
            assert np.array_equal(a[k], b[k]), k
        ids = np.arange(len(names_b.qnames), dtype=np.uint32)
        assert names_a.qnames.take(ids) == names_b.qnames.take(ids)
    finally:
        engine.free_reads(rid_a)
        engine.free_reads(rid_b)
