"""The staged pass of the read stage as a pipeline over chunks of DNMs -- the product's route for a batch larger than one chunk
(hostpath.run_read_phasing) and the thing bench.py measures.

Reference seam: snv_phaser.py:206-299 / sv_phaser.py:176-266 run one task per DNM on a thread pool; each task opens the sites
file and the alignment file for its own window.  Here a batch is cut into chunks (shard.chunk_plan) and every chunk goes through
    site stage   its windows' site + genotype columns up (copy stream), K1 + K2, het lists back: they tell the decoder which fetches the
                 read stage will make -- the ONE host round trip of a chunk
    records      the records those fetches return (+ mates), in the link form, queued on the copy stream
    read stage   queued on the compute stream (uz_phase_begin), picked up later (uz_phase_end)
    (config 5)   allele balance of the chunk's events on its own windows, merged with the read-backed counts
with the stages of neighbouring chunks overlapped: the site windows travel two chunks ahead of their find, the records of chunk k are
enqueued as soon as its het lists are back, and its read stage is queued two finds later -- by then its copy and its header build have run
beside the read stages of the chunks before it; the results are taken one iteration after that.

A chunk is a dict:
    a, b        its DNMs are [a, b) of the batch (results are written there)
    dnms        abi.dnms_view of them
    sites       (held sites view, host site columns incl. "gt", genotype columns {"rd", "ad", "gq"} (8- or 16-bit), wide list or None), all in
                pinned memory -- or None when the caller's family handle `fid` covers the chunk (one site table for the whole batch)
    records     the chunk's records in the link form (abi.Held of uz_reads_packed_view, pinned) -- or a callable returning it, called
                when the chunk's het lists are known: records(k, het_off, het_idx) -> abi.Held   (a decoder working from files)
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def run_pipelined(eng, P, mode, n: int, chunks: Sequence[dict], cnv: bool = False, fid=None, trace: Optional[list] = None, lag: Optional[int] = None,
                  on_done: Optional[Callable] = None, env_lag: bool = True) -> Dict[str, np.ndarray]:
    """One pass over the batch -> per-DNM status / counts / origin / evidence (+ etype / cnv_counts for cnv).
    on_done(k, results of chunk k): called when the read stage of chunk k has been collected and before the next one is queued -- the moment its
    vote lists can still be fetched (uz_phase_votes).  lag: read stages queued this many finds behind (None: from the chunk size); a caller that
    stages chunk k + 1 while chunk k is on the device needs lag + 2 sets of staging buffers."""
    out = dict(status=np.empty(n, np.int32), counts=np.empty((n, 4), np.int32), origin=np.empty(n, np.int32), evidence=np.empty(n, np.int32))
    if cnv:
        out.update(etype=np.empty(n, np.int32), cnv_counts=np.empty((n, 2), np.int32))
    K = len(chunks)
    if K == 0:
        return out
    sids, fids, rids = [None] * K, [fid] * K, [None] * K
    finds = [None] * K
    own_sites = [c.get("sites") is not None for c in chunks]

    def site_stage(k):  # queued on the copy stream (in front of the records of chunk k - 1 ... k): no host wait
        if not own_sites[k]:
            return
        held, hs_k, hg_k, wide_k = chunks[k]["sites"]
        sids[k], fids[k] = eng.upload_sites_family_async(held, hs_k["gt"], hg_k["rd"], hg_k["ad"], hg_k["gq"], wide_k)

    try:
        import inspect
        _transient = "transient" in inspect.signature(eng.find).parameters
    except (TypeError, ValueError):
        _transient = False

    def find_of(k):  # K1 + K2 + the het lists back on the host (read at once by the chunk's decoder: into re-used page-locked buffers where the engine offers them)
        if _transient:
            return eng.find(fids[k], chunks[k]["dnms"], P, mode, transient=True)
        return eng.find(fids[k], chunks[k]["dnms"], P, mode)

    def read_stage_begin(k):  # queued on the compute stream (waits there for the chunk's records): no host wait
        eng.phase_begin(fids[k], rids[k], chunks[k]["dnms"], P, mode)

    pending = {}

    def read_stage_end(k):
        pending[k] = eng.phase_end(fids[k], rids[k], chunks[k]["dnms"], P, mode)
        if on_done is not None:
            on_done(k, pending[k])
        if not cnv:
            finish(k)

    def cnv_stage(k):  # config 5: K6 over the chunk's events, merged with their read-backed counts as summarize_record merges them
        rr = pending[k]
        kk = eng.phase_cnv(fids[k], chunks[k]["dnms"], P, rb_counts=rr["counts"], want_lists=False)
        pending[k] = dict(status=rr["status"], counts=rr["counts"], origin=kk["origin"], evidence=kk["evidence"], etype=kk["etype"], cnv_counts=kk["cnv_counts"])
        finish(k)

    def finish(k):
        a, b = chunks[k]["a"], chunks[k]["b"]
        rr = pending.pop(k)
        for key in out:
            out[key][a:b] = rr[key]
        eng.free_reads(rids[k])
        if own_sites[k]:
            eng.free_sites(sids[k])

    tr = [time.perf_counter()] if trace is not None else None

    def tick():
        if tr is not None:
            tr.append(time.perf_counter())
    # Every wait of the host (the het lists of a find, the results of a read stage) sits behind whatever was queued on the compute stream
    # before it, so the host runs LAG chunks ahead of the read stages: chunk k's records are enqueued as soon as find(k) returns, and its
    # read stage is queued LAG iterations later -- by then the copy (0.8 ms for 12.5 k DNMs) and the header build on its own stream
    # (0.85 ms) have run beside the read stages of the chunks before.  With a lag of one the read stage of chunk k had to wait for that
    # chain (1.65 ms) from the moment the read stage of chunk k - 2 ended: 0.45 ms of idle device per chunk (rocprofv3 timeline of round
    # 4: scripts/staged_timeline.sh).  The site windows travel two chunks ahead of their find, so the link does not idle through that
    # round trip either.  The library keeps the window lists of the last three finds: a lag of two is what it serves.
    # (config 5: the allele-balance stage of a chunk is queued behind the read stage of the next one, and waited for there.)
    if lag is None:
        # two finds ahead pays when a chunk's copy + header build is long (100 k DNMs in 8 chunks of 12.5 k: 12.27 -> 11.9 ms; config 5's three heavy
        # chunks: 4.6 -> 4.5 ms); the three small chunks of a 12.5 k-DNM shard only start their first read stage later (2.45 -> 2.7 ms)
        lag = 2 if (cnv or n >= 8000 * max(1, K)) else 1
    if env_lag:  # (a caller that counted its staging buffers from the lag it passed switches the override off)
        lag = int(os.environ.get("UZ_PIPE_LAG", lag))  # (development aid)
    lag = max(0, min(int(lag), 2, K - 1)) if K > 1 else 1
    site_stage(0)
    if K > 1:
        site_stage(1)
    tick()
    if lag == 0:
        # No lag: the read stage of chunk k is queued the moment its records are enqueued, and find(k + 1) waits behind it on the compute stream --
        # copies and kernels of neighbouring chunks do not overlap, but nothing stands between a chunk's records and its read stage.  For a batch
        # of two small chunks (a 12.5 k-DNM shard of an 8-GPU run) that chain is the shorter one.
        for k in range(K):
            finds[k] = find_of(k)
            tick()
            if k >= 1:
                read_stage_end(k - 1)  # (it ran in front of find(k): no wait)
                if cnv:
                    cnv_stage(k - 1)
            tick()
            rec = chunks[k]["records"]
            if callable(rec):
                rec = rec(k, finds[k][3], finds[k][4])
            finds[k] = None
            rids[k] = eng.upload_reads_packed(rec)
            if k + 2 < K:
                site_stage(k + 2)
            tick()
            read_stage_begin(k)
            tick()
        read_stage_end(K - 1)
        if cnv:
            cnv_stage(K - 1)
        tick()
        if tr is not None:
            trace.append([round((tr[i + 1] - tr[i]) * 1e3, 2) for i in range(len(tr) - 1)])
        return out
    for k in range(K + lag + 1):
        if k < K:
            finds[k] = find_of(k)  # K1 + K2 + the het lists back on the host: the decoder's input
        tick()
        e, b = k - lag - 1, k - lag
        if 0 <= e < K:
            read_stage_end(e)
        tick()
        if 0 <= b < K:
            read_stage_begin(b)
        tick()
        if k < K:
            rec = chunks[k]["records"]
            if callable(rec):
                rec = rec(k, finds[k][3], finds[k][4])
            finds[k] = None
            rids[k] = eng.upload_reads_packed(rec)
            if k + 2 < K:
                site_stage(k + 2)
        if cnv and 0 <= e < K:
            cnv_stage(e)
        tick()
    if tr is not None:
        trace.append([round((tr[i + 1] - tr[i]) * 1e3, 2) for i in range(len(tr) - 1)])
    return out
