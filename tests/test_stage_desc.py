"""The descriptor route of the BAM stage (include/uz_bamwalk.h) on the CPU: the walk's descriptors come from the host's twin of the device kernel
(uz_stage_walk_host), the batch-wide joins (mates, names numbered by first appearance) run on them, and the list of kept records must be the one
the one-pass stage (uz_bam_stage_plan) keeps for the same fetches: same records (by virtual offset), same order, same name ids, same mates, bases
for the same records.  Reference seam: read_collector.py:385, :167 (fetch), :400, :185 (mate)."""
import numpy as np
import pytest

from unfazed_amd import io_native
from test_io_stage import fetches_of


def both(bam, fc, flo, fhi, fex, all_bases=False, threads=3, small_tasks=False):
    src = io_native.BamSource(bam, threads=threads)
    ref = src.select(fc, flo, fhi, 20, extra=fex, all_bases=all_bases)
    n = int(ref.view.n_segs)
    want = io_native.stage_kept_debug(src.lib, ref._stage.ptr, n)
    got = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, small_tasks=small_tasks)
    return ref, want, got, src


def check(ref, want, got):
    voff, qn, mt, bs = want
    assert got.n == voff.size
    d = got.desc
    assert np.array_equal(np.sort(d["src"]), d["src"][np.argsort(d["src"], kind="stable")])
    # kept records -> virtual offsets through the descriptors (a record the host walked itself lies in the aux bytes: its position is read there)
    k = got.kept
    in_aux = (k["src"] & np.uint64(io_native.WALK_SRC_AUX)) != 0
    order = np.argsort(d["src"], kind="stable")
    at = np.searchsorted(d["src"][order], k["src"][~in_aux])
    assert (d["src"][order][at] == k["src"][~in_aux]).all()
    assert np.array_equal(d["voff"][order][at], voff[~in_aux])
    assert np.array_equal(k["qname"], qn)
    assert np.array_equal(k["mate"], mt)
    assert np.array_equal(k["seq_off"] != io_native.KEPT_NO_SEQ, bs.astype(bool))
    assert got.n_qnames == int(ref.view.n_qnames)
    assert np.array_equal(got.contig_off, ref.arrays["contig_off"][: got.contig_off.size])
    assert np.array_equal(got.max_span[: got.n_contigs], ref.arrays["max_span"][: got.n_contigs])
    # offsets: running sums of the records' own sizes
    if got.n:
        assert k["cig_off"][0] == 0 and k["unit_off"][0] == 0 and (np.diff(k["cig_off"].astype(np.int64)) >= 0).all()
        assert got.n_row_units >= got.n_seq_units > 0
    return in_aux


@pytest.mark.parametrize("small_tasks", [False, True])
@pytest.mark.parametrize("stride,spread", [(1, 5), (3, 9), (7, 0)])
def test_kept_list_equals_the_one_pass_stage(workload, stride, spread, small_tasks):
    """small_tasks: the walk plan the device gets -- every task of the stage cut into sub-tasks of ~32 kb of reach, each starting at the record the
    file's linear index names for its first window; their descriptors joined per task (uz_stage_merge_subtasks) must be the whole task's"""
    fc, flo, fhi, fex = fetches_of(workload, stride, spread)
    ref, want, got, _ = both(workload["bam"], fc, flo, fhi, fex, small_tasks=small_tasks)
    in_aux = check(ref, want, got)
    assert not in_aux.any() and got.host_tasks == 0
    if small_tasks:
        assert got.plan["task"].shape[0] >= got.d_first.size - 1 and (stride > 1 or got.plan["task"].shape[0] > got.d_first.size - 1)  # more walk tasks than tasks of the stage
        # (records_walked differs: a sub-task skips the stretch in front of its first window and may share one with its neighbour)
    else:
        assert got.io_stats["records_walked"] == ref.io_stats["records_walked"]


def test_all_bases(workload):
    fc, flo, fhi, fex = fetches_of(workload, 2, 3)
    ref, want, got, _ = both(workload["bam"], fc, flo, fhi, fex, all_bases=True)
    check(ref, want, got)
    assert (got.kept["seq_off"] != io_native.KEPT_NO_SEQ).all()


@pytest.mark.parametrize("slack", ["40", "300"])
def test_mates_through_the_index_travel_as_aux_bytes(workload, slack, monkeypatch):
    fc, flo, fhi, fex = fetches_of(workload, 5, 2)  # (every look-up through the index is a walk of its own: a small batch keeps this on the CPU suite's budget)
    monkeypatch.setenv("UZ_STAGE_SLACK", slack)
    ref, want, got, _ = both(workload["bam"], fc, flo, fhi, fex)
    assert ref.io_stats["index_mate_lookups"] > 0 and got.io_stats["index_mate_lookups"] == ref.io_stats["index_mate_lookups"]
    in_aux = check(ref, want, got)
    assert in_aux.any() and got.n_aux > 0
    # an aux record's bytes are the record: block_size, then refID / pos ... (its position must be the kept record's)
    k = got.kept[in_aux]
    for src in k["src"][:50]:
        o = int(src & np.uint64((1 << 63) - 1))
        bs = int(got.aux[o - 4: o].view("<i4")[0])
        assert 32 <= bs and o + bs <= got.n_aux


def test_a_task_the_device_flags_is_walked_by_the_host(workload):
    """d_flags: every third task handed back as `incomplete` -- the host walks those itself, their records travel as aux bytes, the list is the same"""
    fc, flo, fhi, fex = fetches_of(workload, 1, 5)
    src = io_native.BamSource(workload["bam"], threads=3)
    ref = src.select(fc, flo, fhi, 20, extra=fex)
    want = io_native.stage_kept_debug(src.lib, ref._stage.ptr, int(ref.view.n_segs))
    plain = src.select_kept(fc, flo, fhi, 20, small_tasks=False)

    def walk(plan):
        nt = plan["task"].shape[0]
        flags = np.zeros(nt, np.int32)
        flags[::3] = 1
        keep = np.ones(plain.desc.size, bool)
        for t in range(0, nt, 3):
            keep[plain.d_first[t]: plain.d_first[t + 1]] = False
        cnt = np.diff(plain.d_first).copy()
        cnt[::3] = 0
        d_first = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        return plain.desc[keep].copy(), d_first, flags, np.zeros(nt, np.int64), None

    got = src.select_kept(fc, flo, fhi, 20, walk=walk, small_tasks=False, merge=True)  # (the stage's own tasks as walk tasks: `walk` hands back what `plain` walked)
    in_aux = check(ref, want, got)
    assert in_aux.any() and got.host_tasks > 0


def test_empty_batch(workload):
    z = np.zeros(0, np.int32)
    got = io_native.BamSource(workload["bam"], threads=2).select_kept(z, z, z, 20)
    assert got.n == 0 and got.n_qnames == 0
