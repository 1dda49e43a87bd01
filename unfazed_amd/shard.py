"""Multi-GPU driver: DNMs are independent, so a node phases them as contiguous shards, one
process per GPU, with NO collective on the data path.  The only
communication is the final gather of the per-shard `records` dicts on rank 0 (host objects,
through whatever torch.distributed backend is up: nccl = RCCL on the GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Optional


def shard_bounds(n: int, world: int) -> List[int]:
    """Contiguous, near-equal shards: [b[r], b[r+1]) for rank r."""
    return [(n * r) // world for r in range(world + 1)]


def chunk_plan(n: int, chunks: Optional[int] = None, last_chunk: float = 0.7, min_per_chunk: int = 20000, first_chunk: Optional[float] = None) -> List[int]:
    """Cut points of a staged pass over n DNMs: the uploads of chunk k + 1 overlap the kernels of chunk k.  Every chunk costs a host round trip,
    ~40 kernel launches and the ramp and tail of its own read stage; the first chunk's upload and the last chunk's read stage are what nothing
    hides.  chunks=None: from the shard size -- chunks of at least ~min_per_chunk DNMs, at least three: 100 k DNMs -> five chunks, the last
    last_chunk x the others (round 5, below; round 4 chose seven with a half-size first, round 3 -- link-bound at 7.3 KB per DNM -- eight); a batch
    of fewer than four chunks' worth -> three, the first half-size; a shard below one chunk's size (the 12.5 k DNMs of an 8-GPU run) -> two, the second 1.5 x the first.
    -> [0, ..., n]"""
    if n <= 0:
        return [0, 0]
    # The first chunk half the size of the others (its copy and header build are what nothing hides; with the span sums from the packer the header
    # build is short enough for that to pay: 12.5 k shard 2.45 -> 2.36 ms, config 5 4.63 -> 4.53, 100 k: 6 chunks / 1.0 = 11.45, 6 / 0.5 = 11.69,
    # 7 / 0.5 = 11.07 ms on one box) -- and one chunk more than n / min_per_chunk when there are four or more, so that the others keep their size.
    # Round 5 (read stage one wavefront per DNM, header build from LDS, 4.0 KB per DNM): a timeline of the step shows the compute queue busy 80 % of
    # it and the copies half of it -- what a chunk costs now is the ramp and the tail of its own read stage (a DNM's latency is ~100 us whatever
    # the chunk's size), so fewer, larger chunks win: 4 / 5 / 6 / 7 / 8 chunks with a half-size first = 11.0 / 10.0 / 9.85 / 10.2 / 10.5 ms, and with
    # chunks that large a full-size first one: 5 chunks, first 0.5 / 0.7 / 1.0 = 9.99 / 9.83 / 9.69 ms (last 0.5 / 0.7 / 1.0: 9.85 / 9.69 / 9.70).
    k0 = n // int(min_per_chunk)
    k = int(chunks) if chunks else max(3, k0)
    if not chunks and k0 < 1:
        # a shard below one chunk's size (the 12.5 k DNMs of an 8-GPU run): two chunks -- round 5, with the read stage and the header build at a
        # third of their round-4 time a chunk's fixed costs (a find, ~40 launches) weigh more than what a third chunk hides: 1 / 2 / 3 / 4 chunks =
        # 2.75 / 2.57 / 2.75 / 3.10 ms on one box -- and the FIRST the smaller one, the second 1.5 x it: what nothing hides in so short a pass is the
        # first chunk's way to the device (find, copy, header build), and since the find's answers no longer wait behind the uploads (uz_find: copy
        # kernels) the second chunk's copy is done before the first read stage is: second chunk 0.6 / 1.0 / 1.2 / 1.5 / 1.8 / 2.0 x the first =
        # 2.45 / 2.21 / 2.18 / 2.15 / 2.17 / 2.20 ms
        k, last_chunk = 2, 1.5
    if first_chunk is None:
        first_chunk = 1.0 if (not chunks and k0 >= 4) else 0.5
    k = max(1, min(k, n))
    f = min(3.0, max(0.05, float(last_chunk)))  # (above 1: a last chunk LARGER than the others -- a two-chunk shard whose first chunk should reach the device early)
    g = min(1.0, max(0.05, float(first_chunk))) if k >= 3 else 1.0
    unit = n / (k - 2 + g + f) if k >= 2 else float(n)
    cuts = [0] + [min(n, int(round(unit * (g + j)))) for j in range(k - 1)] + [n] if k >= 2 else [0, n]
    out = [0]
    for c in cuts[1:]:
        if c > out[-1]:
            out.append(c)
    return out


def shard_dnms(dnms: List[dict], rank: int, world: int) -> List[dict]:
    """Sort by (chrom, start, end) -- neighbours share site windows and read blocks; ties keep the input order, as the single
    process meets them -- and take this rank's contiguous slice."""
    order = sorted(range(len(dnms)), key=lambda i: (str(dnms[i]["chrom"]), int(dnms[i]["start"]), int(dnms[i]["end"]), i))
    b = shard_bounds(len(order), world)
    # DNMs that share a chromosome and a start stay in one shard: the many-variant path counts them together (find_many's
    # per-position lists, informative_site_finder.py:385-395)
    same = lambda x, y: (str(dnms[x]["chrom"]), int(dnms[x]["start"])) == (str(dnms[y]["chrom"]), int(dnms[y]["start"]))  # noqa: E731
    for r in range(1, world):
        while 0 < b[r] < len(order) and same(order[b[r] - 1], order[b[r]]):
            b[r] += 1
        b[r] = max(b[r], b[r - 1])
    return [dnms[i] for i in order[b[rank]: b[rank + 1]]]


def phase_sharded(phase_fn, dnms: List[dict], *args, rank: int = 0, world: int = 1, dist=None, **kw) -> Optional[Dict[str, dict]]:
    """Run `phase_fn(shard, *args, **kw)` (phase_snvs / phase_svs) on this rank's shard and gather the
    records on rank 0 (returns None on the other ranks).  With world == 1 this is a plain call."""
    mine = shard_dnms(dnms, rank, world) if world > 1 else dnms
    recs = phase_fn(mine, *args, **kw) if mine else {}
    if world == 1 or dist is None:
        return recs
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(recs, gathered, dst=0)
    if rank != 0:
        return None
    merged: Dict[str, dict] = {}
    for part in gathered:
        merged.update(part)
    return merged
