"""The batch-wide joins as the device formulates them (tests/joinmodel.py: order-free rules over all descriptors of a batch) against the one-pass
stage (uz_bam_stage_plan: the host's task-by-task hash tables, frontiers and stable sort) on the CPU -- same records, same order, same name ids,
same mates, bases for the same records -- and the host's share of the device route: tasks handed back (uz_stage_walk_flagged) and mates looked up
through the index (uz_stage_lookup).  Reference seam: read_collector.py:400, :185 (mate), :226-234 (name-keyed tables)."""
import ctypes as C

import numpy as np
import pytest

import joinmodel
from unfazed_amd import io_native
from test_io_stage import fetches_of


def begin(src, fc, flo, fhi, all_bases=False):
    """a begun stage with its blocks gathered and its walk plan (the stage's own tasks as walk tasks), and the twin's descriptors"""
    lib = src.lib
    st = C.c_void_p()
    flags = io_native.STAGE_ALL_BASES if all_bases else 0
    io_native._check(lib, lib.uz_bam_stage_begin(src._h.ptr, int(fc.size), fc.ctypes.data, flo.ctypes.data, fhi.ctypes.data, None, flags, 20, int(src.threads), C.byref(st)))
    sh = io_native._Handle(st.value, lib.uz_stage_free)
    nb, cb, ob = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    io_native._check(lib, lib.uz_stage_gather_blocks(sh.ptr, None, 0, None, None, C.byref(nb), C.byref(cb), C.byref(ob)))
    comp = np.zeros(int(cb.value) + 64, np.uint8)
    in_off, out_off = np.zeros(max(1, nb.value), np.int64), np.zeros(nb.value + 1, np.int64)
    if nb.value:
        io_native._check(lib, lib.uz_stage_gather_blocks(sh.ptr, comp.ctypes.data, int(cb.value), in_off.ctypes.data, out_off.ctypes.data, None, None, None))
    z = (C.c_int64 * 8)()
    lib.uz_stage_walk_plan_sizes(sh.ptr, z)
    nt, nsp, nr, nf, nblk, n_host = (int(x) for x in z[:6])
    assert nt == n_host
    task = np.zeros((max(1, nt), io_native.WALK_TASK_COLS), np.int32)
    span = np.zeros((max(1, nsp), io_native.WALK_SPAN_COLS), np.int64)
    reach = np.zeros((max(1, nr), 2), np.int32)
    fetch = np.zeros((max(1, nf), 3), np.int32)
    blk_coff = np.zeros(max(1, nblk), np.int64)
    io_native._check(lib, lib.uz_stage_walk_plan(sh.ptr, task.ctypes.data, span.ctypes.data, reach.ctypes.data, fetch.ctypes.data, blk_coff.ctypes.data, None))
    d_first, d_walked = np.zeros(n_host + 1, np.int64), np.zeros(max(1, n_host), np.int64)
    io_native._check(lib, lib.uz_stage_walk_host(sh.ptr, None, 0, d_first.ctypes.data, d_walked.ctypes.data))
    desc = np.zeros(max(1, int(d_first[-1])), io_native.WALK_DESC)
    io_native._check(lib, lib.uz_stage_walk_host(sh.ptr, desc.ctypes.data, int(desc.size), d_first.ctypes.data, d_walked.ctypes.data))
    return sh, dict(task=task[:nt], reach=reach[:nr]), desc[: int(d_first[-1])], d_first


def model_vs_stage(bam, fc, flo, fhi, fex, all_bases=False, flag_every=0):
    src = io_native.BamSource(bam, threads=3)
    ref = src.select(fc, flo, fhi, 20, extra=fex, all_bases=all_bases)
    voff, qn, mt, bs = io_native.stage_kept_debug(src.lib, ref._stage.ptr, int(ref.view.n_segs))
    sh, plan, desc, d_first = begin(src, fc, flo, fhi, all_bases)
    nt = plan["task"].shape[0]
    d_flags = np.zeros(max(1, nt), np.int32)
    if flag_every:  # the device hands tasks back: their descriptors are void
        d_flags[::flag_every] = 1
    got = joinmodel.run(src.lib, sh.ptr, desc, d_first, d_flags, plan, len(src.contigs), all_bases=all_bases)
    assert np.array_equal(got["voff"], voff)
    assert np.array_equal(got["qname"], qn)
    assert np.array_equal(got["mate"], mt)
    assert np.array_equal(got["bases"], bs.astype(bool))
    assert got["n_qnames"] == int(ref.view.n_qnames)
    return ref, got


@pytest.mark.parametrize("stride,spread", [(1, 5), (3, 9), (7, 0)])
def test_the_order_free_joins_equal_the_one_pass_stage(workload, stride, spread):
    fc, flo, fhi, fex = fetches_of(workload, stride, spread)
    ref, got = model_vs_stage(workload["bam"], fc, flo, fhi, fex)
    assert got["lookups"] == 0 and got["n_extra"] == 0


def test_all_bases(workload):
    fc, flo, fhi, fex = fetches_of(workload, 2, 3)
    model_vs_stage(workload["bam"], fc, flo, fhi, fex, all_bases=True)


@pytest.mark.parametrize("slack", ["0", "300"])
def test_mates_through_the_index(workload, slack, monkeypatch):
    """a small reach slack pushes the mates out of the reach intervals: they come back from uz_stage_lookup as descriptors in the aux store, and the
    closure goes on from them"""
    fc, flo, fhi, fex = fetches_of(workload, 5, 2)
    monkeypatch.setenv("UZ_STAGE_SLACK", slack)
    ref, got = model_vs_stage(workload["bam"], fc, flo, fhi, fex)
    assert got["lookups"] > 0 and got["n_extra"] > 0 and got["trips"] <= 4  # (trips through the index: fetched records, their mates' copies -- then nobody asks)
    assert ((got["src"] & np.uint64(io_native.WALK_SRC_AUX)) != 0).any()


def test_tasks_the_device_hands_back(workload):
    fc, flo, fhi, fex = fetches_of(workload, 1, 5)
    ref, got = model_vs_stage(workload["bam"], fc, flo, fhi, fex, flag_every=3)
    assert got["h_flags"].any() and got["n_extra"] > 0


def test_tasks_handed_back_and_mates_through_the_index(workload, monkeypatch):
    monkeypatch.setenv("UZ_STAGE_SLACK", "60")
    fc, flo, fhi, fex = fetches_of(workload, 5, 3)  # (the model is plain Python: a small batch)
    ref, got = model_vs_stage(workload["bam"], fc, flo, fhi, fex, flag_every=2)
    assert got["lookups"] > 0 and got["h_flags"].any()


def _small_fetches(ds, full, seed, n_het):
    rng = np.random.default_rng(seed)
    c, lo, hi, ex = [], [], [], []
    for d in ds.dnms:
        tid = full.contig_index[d["chrom"]]
        c.append(tid); lo.append(d["start"] - 1); hi.append(d["start"] + 1); ex.append(max(1, d["end"] - d["start"]))
        for p in np.sort(rng.integers(d["start"] - 5000, d["start"] + 5000, n_het)):
            c.append(tid); lo.append(int(p)); hi.append(int(p) + 1); ex.append(0)
    return np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32), np.array(ex, np.uint16)


@pytest.mark.parametrize("seed,readlen", [(33, 301)])
def test_odd_records_from_a_python_written_bam(tmp_path, seed, readlen, monkeypatch):
    """synth.small's pile-ups written by the Python BAM writer: duplicates, secondary / supplementary copies with SA tags, mates unmapped or on other
    contigs, overlapping mates -- mate() on a name with more than two records, names that share no pair"""
    from filesio import dump_dataset, write_bai
    from synth.small import SmallConfig, make_small
    ds = make_small(SmallConfig(seed=seed, n_dnms=10, odd_read_prob=0.25, lowq_prob=0.08, softclip_prob=0.05, indel_prob=0.03, readlen=readlen, cluster_prob=0.6))
    bam = list(dump_dataset(ds, str(tmp_path))["bams"].values())[0]
    write_bai(bam)
    full = io_native.read_bam_table(bam, threads=2)
    fc, flo, fhi, fex = _small_fetches(ds, full, seed, 8)
    model_vs_stage(bam, fc, flo, fhi, fex)
    monkeypatch.setenv("UZ_STAGE_SLACK", "30")
    ref, got = model_vs_stage(bam, fc, flo, fhi, fex, flag_every=4)
    assert got["lookups"] > 0


def test_reads_longer_than_the_reach_slack(tmp_path):
    """2.5 kb reads: a record overlaps the fetches of two tasks -- both keep it, one copy survives (the one a fetch returned), and the mates lie
    beyond any slack"""
    from filesio import dump_dataset, write_bai
    from synth.small import SmallConfig, make_small
    ds = make_small(SmallConfig(seed=41, n_dnms=6, odd_read_prob=0.1, readlen=2500, coverage_per_hap=4.0, ins_mean=7500, ins_sd=100))
    bam = list(dump_dataset(ds, str(tmp_path))["bams"].values())[0]
    write_bai(bam)
    full = io_native.read_bam_table(bam, threads=2)
    fc, flo, fhi, fex = _small_fetches(ds, full, 5, 3)
    ref, got = model_vs_stage(bam, fc, flo, fhi, fex)
    assert got["lookups"] > 0
