"""The order in which pipeline.run_pipelined drives the C ABI, on a stand-in engine (no GPU): the contract of unfazed_hip.h must hold for every
chunk count and lag -- one read stage open at a time (uz_phase_begin / uz_phase_end), a read stage queued only while the window lists of its own
find are still among the last three the library keeps, every chunk's records uploaded after its find and before its read stage, site windows in
place before their find, vote lists fetched between a read stage's end and the next one's begin, results written to the right rows."""
import numpy as np
import pytest

from unfazed_amd import pipeline


class Engine:
    def __init__(self, rows=2):
        self.log, self.open, self.finds, self.uploaded, self.sites_up, self.rows = [], None, [], set(), 0, rows

    def upload_sites_family_async(self, held, gt, rd, ad, gq, wide):
        self.sites_up += 1
        self.log.append(("S", self.sites_up - 1))
        return 1000 + self.sites_up, 2000 + self.sites_up

    def find(self, fid, d, P, mode):
        assert fid is not None
        self.log.append(("f", d))
        self.finds.append(d)
        return (0, 0, 0, np.array([0, 1]), np.array([d]))

    def phase_begin(self, fid, rid, d, P, mode):
        assert self.open is None, "a read stage is still open"
        assert d in self.finds[-3:], "the window lists of this chunk's find are gone (the library keeps three)"
        assert rid in self.uploaded
        self.open = d
        self.log.append(("b", d))

    def phase_end(self, fid, rid, d, P, mode):
        assert self.open == d
        self.open = None
        self.log.append(("e", d))
        r = self.rows
        return dict(status=np.full(r, d, np.int32), counts=np.full((r, 4), d, np.int32), origin=np.full(r, d, np.int32), evidence=np.full(r, d, np.int32))

    def phase_cnv(self, fid, d, P, rb_counts=None, want_lists=False):
        self.log.append(("c", d))
        r = self.rows
        return dict(origin=np.full(r, 10 + d, np.int32), evidence=np.full(r, d, np.int32), etype=np.full(r, d, np.int32), cnv_counts=np.full((r, 2), d, np.int32))

    def upload_reads_packed(self, rec):
        self.log.append(("u", rec))
        self.uploaded.add(("rid", rec))
        return ("rid", rec)

    def free_reads(self, rid):
        self.uploaded.discard(rid)

    def free_sites(self, sid):
        pass


def _chunks(K, own_sites=True, lazy=False, seen=None):
    def rec_of(k):
        if not lazy:
            return k

        def make(kk, ho, hi):
            assert int(hi[0]) == kk  # the het lists handed over are this chunk's own
            seen.append(kk)
            return kk
        return make
    return [dict(a=2 * k, b=2 * k + 2, dnms=k, records=rec_of(k), sites=(0, {"gt": 0}, {"rd": 0, "ad": 0, "gq": 0}, None) if own_sites else None) for k in range(K)]


@pytest.mark.parametrize("K", [1, 2, 3, 4, 6, 9])
@pytest.mark.parametrize("lag", [None, 1, 2])
@pytest.mark.parametrize("cnv", [False, True])
def test_order_of_calls(K, lag, cnv, monkeypatch):
    monkeypatch.delenv("UZ_PIPE_LAG", raising=False)
    e = Engine()
    seen, hooks = [], []
    out = pipeline.run_pipelined(e, None, 0, 2 * K, _chunks(K, lazy=True, seen=seen), cnv=cnv, lag=lag, on_done=lambda k, rr: hooks.append((k, e.open)))
    pos = {x: i for i, x in enumerate(e.log)}
    for k in range(K):
        assert pos[("S", k)] < pos[("f", k)] < pos[("u", k)] < pos[("b", k)] < pos[("e", k)]
        if k:
            assert pos[("e", k - 1)] < pos[("b", k)] and pos[("f", k - 1)] < pos[("f", k)]
        if cnv:
            assert pos[("e", k)] < pos[("c", k)]
    assert seen == list(range(K)) and e.open is None and not e.uploaded
    assert hooks == [(k, None) for k in range(K)]  # (between a read stage's end and the next begin: its vote lists are still there)
    want = np.repeat(np.arange(K), 2)
    assert np.array_equal(out["status"], want) and np.array_equal(out["counts"][:, 3], want)
    assert np.array_equal(out["origin"], want + 10 if cnv else want)
    # the lag really is what was asked for (up to what the chunk count allows): finds issued before the first read stage is queued
    if lag is not None and K > 1:
        eff = min(lag, 2, K - 1)
        assert sum(1 for x in e.log[: pos[("b", 0)]] if x[0] == "f") == eff + 1


def test_one_site_table_for_the_whole_batch_and_the_default_lag():
    e = Engine()
    out = pipeline.run_pipelined(e, None, 0, 6, _chunks(3, own_sites=False), fid=7)
    assert e.sites_up == 0 and [x for x in e.log if x[0] == "f"] == [("f", 0), ("f", 1), ("f", 2)]
    assert np.array_equal(out["evidence"], [0, 0, 1, 1, 2, 2])
    # small chunks: one find ahead; chunks of 8 k DNMs and more (or heavy SV chunks): two
    for n, K, cnv, want in ((12500, 3, False, 1), (100000, 6, False, 2), (10000, 3, True, 2)):
        e = Engine(rows=0)
        ch = [dict(a=0, b=0, dnms=k, records=k, sites=None) for k in range(K)]
        pipeline.run_pipelined(e, None, 0, n, ch, cnv=cnv, fid=7)
        first_b = e.log.index(("b", 0))
        assert sum(1 for x in e.log[:first_b] if x[0] == "f") == want + 1


def test_a_chunks_result_lists_as_strings_in_one_pass_equal_the_lists_built_entry_by_entry():
    """hostpath._VoteLists.strings (a fancy index over the distinct names, one str() per distinct position) against the per-DNM path of
    _evidence_lists: the same four lists for every phased DNM, whichever way they are built; the cyclic collector comes back on after a
    phasing call that switched it off (session.no_gc_pauses)."""
    import gc
    import numpy as np
    from unfazed_amd import abi, session
    from unfazed_amd.hostpath import PhasingHost, _VoteLists
    rng = np.random.default_rng(12)
    n = 200
    lens = rng.integers(0, 9, 4 * n)
    lens[rng.random(4 * n) < 0.3] = 0
    vo = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    vv = np.zeros(int(vo[-1]), np.int32)
    for k in range(4 * n):
        a, b = int(vo[k]), int(vo[k + 1])
        vv[a:b] = rng.integers(0, 500, b - a) if k % 4 < 2 else rng.integers(1, 10 ** 8, b - a)  # read ids / site positions
    status = np.where(rng.random(n) < 0.8, abi.ST_OK, abi.ST_NO_OVERLAP).astype(np.int32)

    class Names:
        calls = 0

        def take(self, ids):
            Names.calls += 1
            return ["read%05d" % int(i) for i in ids]

        def __getitem__(self, i):
            return "read%05d" % int(i)

    class Table:
        qnames = Names()

    res = dict(status=status, lists=_VoteLists(vo, vv))
    fast = PhasingHost._evidence_lists(res, range(n), Table())
    assert Names.calls == 1  # one look-up for the whole chunk
    slow_res = dict(status=status, lists=[tuple(vv[vo[4 * k + q]: vo[4 * k + q + 1]] for q in range(4)) for k in range(n)])
    slow = PhasingHost._evidence_lists(slow_res, range(n), Table())
    assert set(fast) == set(slow) == {k for k in range(n) if status[k] == abi.ST_OK}
    for k in fast:
        assert tuple(fast[k]) == tuple(slow[k]), k
        assert all(isinstance(x, str) for part in fast[k] for x in part)
    # a few DNMs of the chunk only: the per-DNM path of the same object
    some = PhasingHost._evidence_lists(dict(status=status, lists=_VoteLists(vo, vv)), [k for k in range(n) if k % 7 == 0], Table())
    for k in some:
        assert tuple(some[k]) == tuple(slow[k])
    was = gc.isenabled()
    gc.enable()
    with session.no_gc_pauses():
        assert not gc.isenabled()
    assert gc.isenabled()
    gc.disable()
    with session.no_gc_pauses():
        assert not gc.isenabled()
    assert not gc.isenabled()  # (it was off before the call: it stays off)
    if was:
        gc.enable()


def test_the_gc_pause_is_counted_across_threads():
    """session.no_gc_pauses: the collector pauses while any phasing call is in flight and comes back with the last one out, whatever the threads'
    interleaving; a caller who had it off keeps it off"""
    import gc
    import threading
    from unfazed_amd import session
    assert gc.isenabled()
    inside, go, seen = threading.Barrier(3), threading.Event(), []

    def call(hold):
        with session.no_gc_pauses():
            seen.append(gc.isenabled())
            inside.wait()
            if hold:
                go.wait()
        seen.append(("out", gc.isenabled()))

    ts = [threading.Thread(target=call, args=(h,)) for h in (False, True)]
    for t in ts:
        t.start()
    inside.wait()          # both are inside
    ts[0].join()           # the first one leaves: the other is still phasing
    assert not gc.isenabled()
    go.set()
    ts[1].join()
    assert gc.isenabled() and seen[:2] == [False, False]
    gc.disable()
    try:
        with session.no_gc_pauses():
            assert not gc.isenabled()
        assert not gc.isenabled()  # the caller's own setting is what is left behind
    finally:
        gc.enable()
