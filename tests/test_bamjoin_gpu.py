"""The batch-wide joins of the BAM stage on the device (csrc/k_bamjoin.hip, uz_bam_join) through the C ABI -- mate() for every fetched read and every
mate of a mate (read_collector.py:400, :185), names numbered by first appearance (the name-keyed tables of read_collector.py:226-234):
  * the kept list the device builds in HBM == the host's (plan_finish over the same walk's descriptors: uz_bam_stage_finish_sub) byte for byte --
    name ids, mates, the offsets into the CIGAR / quality-row / base-row / name stores, bases for the same records -- and == the one-pass stage;
  * with mates only the index can answer (a small reach slack) and with walk tasks handed back to the host: the host contributes descriptors,
    the closure still runs on the device;
  * the table built from that list == the table the host route stages, header by header; read names by id;
  * the slots of walked batches: a process whose batches stopped growing allocates nothing more."""
import numpy as np
import pytest

from unfazed_amd import io_native
from test_io_stage import fetches_of

pytestmark = pytest.mark.gpu


def joined_vs_host(engine, bam, fc, flo, fhi, fex, all_bases=False, want_lookups=False, want_flagged=False):
    src = io_native.BamSource(bam, threads=3)
    ref = src.select(fc, flo, fhi, 20, extra=fex, all_bases=all_bases)  # the one-pass stage
    n = int(ref.view.n_segs)
    voff, qn, mt, bs = io_native.stage_kept_debug(src.lib, ref._stage.ptr, n)
    host = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, walk=engine.bam_walk, release=engine.bam_walk_release)  # the device's walk, the HOST's joins
    dev = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, join=engine, release=engine.bam_walk_release)            # ... and the DEVICE's joins
    try:
        assert dev.joined and dev.n == n == host.n
        got = engine.join_fetch(dev.token, dev.n, len(src.contigs))
        assert np.array_equal(got["voff"], voff)
        assert np.array_equal(got["qname"], qn) and np.array_equal(got["mate"], mt) and np.array_equal(got["bases"], bs)
        for f in ("qname", "mate", "cig_off", "unit_off", "seq_off", "name_off"):
            assert np.array_equal(got["kept"][f], host.kept[f]), f
        in_hbm = ((got["kept"]["src"] | host.kept["src"]) & np.uint64(io_native.WALK_SRC_AUX)) == 0
        assert np.array_equal(got["kept"]["src"][in_hbm], host.kept["src"][in_hbm])
        assert (dev.n_qnames, dev.n_cigar_total, dev.n_row_units, dev.n_seq_units, dev.n_name_bytes) == \
            (host.n_qnames, host.n_cigar_total, host.n_row_units, host.n_seq_units, host.n_name_bytes)
        assert np.array_equal(got["contig_off"], host.contig_off) and np.array_equal(got["max_span"][: host.n_contigs], host.max_span[: host.n_contigs])
        if want_lookups:
            assert dev.io_stats["index_mate_lookups"] > 0 and dev.join_calls > 1 and dev.n_aux > 0
        if want_flagged:
            assert dev.host_tasks > 0 and dev.n_aux > 0
    finally:
        engine.bam_walk_release(dev.token)
        dev.token = None
        del host
    return ref, dev


@pytest.mark.parametrize("stride,spread", [(1, 5), (3, 9), (7, 0)])
def test_kept_list_in_hbm_equals_the_hosts(engine, workload, stride, spread):
    fc, flo, fhi, fex = fetches_of(workload, stride, spread)
    ref, dev = joined_vs_host(engine, workload["bam"], fc, flo, fhi, fex)
    assert dev.join_calls == 1 and dev.n_extra_desc == 0  # nothing of this batch needed the host


def test_all_bases(engine, workload):
    fc, flo, fhi, fex = fetches_of(workload, 2, 3)
    joined_vs_host(engine, workload["bam"], fc, flo, fhi, fex, all_bases=True)


@pytest.mark.parametrize("slack", ["0", "40", "300"])
def test_mates_only_the_index_can_answer(engine, workload, slack, monkeypatch):
    fc, flo, fhi, fex = fetches_of(workload, 5, 2)
    monkeypatch.setenv("UZ_STAGE_SLACK", slack)
    joined_vs_host(engine, workload["bam"], fc, flo, fhi, fex, want_lookups=True)


def test_walk_tasks_handed_back_to_the_host(engine, workload, monkeypatch):
    fc, flo, fhi, fex = fetches_of(workload, 1, 5)
    monkeypatch.setenv("UZ_TEST_FLAG_EVERY", "5")
    joined_vs_host(engine, workload["bam"], fc, flo, fhi, fex, want_flagged=True)
    monkeypatch.setenv("UZ_STAGE_SLACK", "60")
    joined_vs_host(engine, workload["bam"], fc, flo, fhi, fex, want_flagged=True, want_lookups=True)


def _small_fetches(ds, full, seed, n_het):
    rng = np.random.default_rng(seed)
    c, lo, hi, ex = [], [], [], []
    for d in ds.dnms:
        tid = full.contig_index[d["chrom"]]
        c.append(tid); lo.append(d["start"] - 1); hi.append(d["start"] + 1); ex.append(max(1, d["end"] - d["start"]))
        for p in np.sort(rng.integers(d["start"] - 5000, d["start"] + 5000, n_het)):
            c.append(tid); lo.append(int(p)); hi.append(int(p) + 1); ex.append(0)
    return np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32), np.array(ex, np.uint16)


@pytest.mark.parametrize("seed,readlen", [(31, 151), (32, 100), (33, 301)])
def test_odd_records_from_a_python_written_bam(engine, tmp_path, seed, readlen, monkeypatch):
    """duplicates, secondary / supplementary copies with SA tags, mates unmapped or on other contigs, overlapping mates: names with more than two
    records, names whose mate nobody holds"""
    from filesio import dump_dataset, write_bai
    from synth.small import SmallConfig, make_small
    ds = make_small(SmallConfig(seed=seed, n_dnms=10, odd_read_prob=0.25, lowq_prob=0.08, softclip_prob=0.05, indel_prob=0.03, readlen=readlen, cluster_prob=0.6))
    bam = list(dump_dataset(ds, str(tmp_path))["bams"].values())[0]
    write_bai(bam)
    full = io_native.read_bam_table(bam, threads=2)
    fc, flo, fhi, fex = _small_fetches(ds, full, seed, 8)
    joined_vs_host(engine, bam, fc, flo, fhi, fex)
    monkeypatch.setenv("UZ_STAGE_SLACK", "30")
    monkeypatch.setenv("UZ_TEST_FLAG_EVERY", "4")
    joined_vs_host(engine, bam, fc, flo, fhi, fex, want_lookups=True)


def test_reads_longer_than_the_reach_slack(engine, tmp_path):
    """2.5 kb reads: a record overlaps the fetches of two tasks, both keep it, one copy survives; the kept records of neighbouring tasks interleave"""
    from filesio import dump_dataset, write_bai
    from synth.small import SmallConfig, make_small
    ds = make_small(SmallConfig(seed=41, n_dnms=6, odd_read_prob=0.1, readlen=2500, coverage_per_hap=4.0, ins_mean=7500, ins_sd=100))
    bam = list(dump_dataset(ds, str(tmp_path))["bams"].values())[0]
    write_bai(bam)
    full = io_native.read_bam_table(bam, threads=2)
    fc, flo, fhi, fex = _small_fetches(ds, full, 5, 3)
    joined_vs_host(engine, bam, fc, flo, fhi, fex, want_lookups=True)


@pytest.mark.parametrize("all_bases,slack", [(False, None), (True, None), (False, "40")])
def test_table_built_from_the_list_in_hbm(engine, workload, all_bases, slack, monkeypatch):
    """uz_reads_from_walk: the record table from the kept list in HBM == the table the host route stages, header by header; names by id from HBM"""
    if slack:
        monkeypatch.setenv("UZ_STAGE_SLACK", slack)
    fc, flo, fhi, fex = fetches_of(workload, 1 if not slack else 2, 5 if not slack else 3)
    src = io_native.BamSource(workload["bam"], threads=3)
    staged = src.select(fc, flo, fhi, 20, extra=fex, all_bases=all_bases)
    rid_a = engine.upload_reads_packed(staged)
    engine.wait_reads(rid_a)
    dev = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, join=engine, release=engine.bam_walk_release)
    rid_b = engine.reads_from_bam(dev, names=True)
    try:
        n = int(staged.view.n_segs)
        assert dev.n == n and dev.token is None
        a, b = engine.reads_headers(rid_a, n), engine.reads_headers(rid_b, n)
        for k in ("start", "end", "tlen", "mate", "qname"):
            assert np.array_equal(a[k], b[k]), k
        ids = np.arange(len(staged.qnames), dtype=np.uint32)
        assert len(dev.qnames) == ids.size
        assert dev.qnames.take(ids) == staged.qnames.take(ids)
        assert dev.qnames[3] == staged.qnames[3] and dev.qnames.take(ids[::-7]) == staged.qnames.take(ids[::-7])
    finally:
        engine.free_reads(rid_a)
        engine.free_reads(rid_b)


def test_empty_batch(engine, workload):
    z = np.zeros(0, np.int32)
    src = io_native.BamSource(workload["bam"], threads=2)
    dev = src.select_kept(z, z, z, 20, join=engine, release=engine.bam_walk_release)
    assert dev.n == 0 and dev.n_qnames == 0
    rid = engine.reads_from_bam(dev)
    engine.free_reads(rid)


def test_slots_grow_once(engine, workload):
    """batches of 1 x / 1.5 x / 1 x the size, again and again: a slot that meets a larger batch grows to the largest sizes the context has seen,
    nothing is freed under running streams, and from the second time round no call allocates (round 5: a process's second call met hipFree +
    hipMalloc of gigabytes with every stream waiting)"""
    src = io_native.BamSource(workload["bam"], threads=3)
    small = fetches_of(workload, 3, 4)
    large = fetches_of(workload, 1, 9)

    def one(f, keep_open=None):
        dev = src.select_kept(f[0], f[1], f[2], 20, join=engine, release=engine.bam_walk_release)
        if keep_open is not None:
            keep_open.append(dev)
            return
        engine.free_reads(engine.reads_from_bam(dev))

    held = []
    for f in (small, large, small):  # three batches in flight at once, on three slots
        one(f, held)
    for dev in held:
        engine.free_reads(engine.reads_from_bam(dev))
    for f in (small, large, small, large):
        one(f)
    before = engine.walk_slot_stats()
    held = []
    for f in (large, small, large):  # now every slot a batch can land on has seen the large one or is grown to it at once
        one(f, held)
    for dev in held:
        engine.free_reads(engine.reads_from_bam(dev))
    mid = engine.walk_slot_stats()
    for f in (small, large, small, large, large, small):
        one(f)
    held = []
    for f in (large, large, small):
        one(f, held)
    for dev in held:
        engine.free_reads(engine.reads_from_bam(dev))
    after = engine.walk_slot_stats()
    assert after["allocations"] == mid["allocations"], (before, mid, after)
