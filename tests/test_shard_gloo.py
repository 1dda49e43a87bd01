"""The N > 1 path on CPU: two gloo ranks phase disjoint DNM shards (host path driven through the
oracle backend) and rank 0's merged records equal the single-process result."""
import json
import os
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import RUN_DEFAULTS, tables
    from oracle_backend import OracleBackend
    from synth.small import SmallConfig, make_small
    from unfazed_amd import session
    from unfazed_amd.shard import phase_sharded
    from unfazed_amd.snv_phaser import phase_snvs
    ds = make_small(SmallConfig(seed=808, n_dnms=9, kids=["kidA", "kidB"]))
    sites, reads = tables(ds)
    session.set_backend(OracleBackend())
    session.register_sites("mem://sites", sites)
    for k, t in reads.items():
        session.register_reads(k, t)
    a = RUN_DEFAULTS
    recs = phase_sharded(
        phase_snvs, ds.dnms, list(ds.pedigrees), ds.pedigrees, "mem://sites", 1, a["build"], False, 1000, True,
        a["ab_homref"], a["ab_homalt"], a["ab_het"], 20, 10, 5000, 1000000, 3, 1, 151, 5,
        rank=rank, world=world, dist=dist)
    if rank == 0:
        json.dump({k: {kk: (sorted(v) if isinstance(v, list) else v) for kk, v in r.items()} for k, r in recs.items()},
                  open(out_path, "w"), sort_keys=True)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, tmp_path, port):
    out = os.path.join(str(tmp_path), "w%d.json" % world)
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    return json.load(open(out))


def test_two_ranks_equal_one_rank(tmp_path):
    one = _run(1, tmp_path, 29731)
    two = _run(2, tmp_path, 29732)
    assert one == two
    assert len(one) >= 2


def test_shard_bounds_cover_everything():
    from unfazed_amd.shard import shard_bounds, shard_dnms
    for n in (0, 1, 7, 8, 100001):
        for w in (1, 2, 4, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
            assert max(b[i + 1] - b[i] for i in range(w)) - min(b[i + 1] - b[i] for i in range(w)) <= 1
    dn = [dict(chrom=str(c), start=s, end=s + 1, kid="k") for c in (2, 1) for s in (5, 3, 9)]
    got = [d for r in range(3) for d in shard_dnms(dn, r, 3)]
    assert sorted((d["chrom"], d["start"]) for d in got) == sorted((d["chrom"], d["start"]) for d in dn)
