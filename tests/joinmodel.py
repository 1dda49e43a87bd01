"""The batch-wide joins of the BAM stage as the DEVICE formulates them (csrc/k_bamjoin.hip), stated in numpy / plain Python -- test infrastructure.

What they replace: `bamfile.mate(read)` per fetched read and per mate of a mate (read_collector.py:400, :185) and the name-keyed tables of
read_collector.py:226-234, which the host's `plan_finish` (csrc/io_stage.cpp) answers task by task with hash tables, a frontier per generation and a
stable sort.  The device cannot walk records one after the other, so its rules are ORDER-FREE statements over all descriptors of a batch:

  * a record's join task: the stage task its walk task belongs to (or what the host says for a record it walked itself);
  * dropped: the copy a later walk sub-task made of a record the one before met (pos < the stop of the sub-task before), and every descriptor of a
    task the host walks itself;
  * mate(x) = among the records of the join task whose reach interval holds x's mate position (or of the look-up task the host made for it) with
    x's name (two hashes + length) and the other read-of-pair flag that overlap the mate position, the one with the smallest virtual offset;
    no record of that name in that task -> the index has to answer (a `need`);
  * members = the fetched records, closed under mate(); kept = members, one per virtual offset (the copy a fetch returned wins);
  * order = virtual offset; name ids = rank of a name's first kept record among the first kept records; mate = the kept index of mate(x)'s survivor.

`tests/test_join_model.py` holds this against the one-pass stage on the CPU; `tests/test_bamjoin_gpu.py` holds the kernels against both."""
import ctypes as C

import numpy as np

from unfazed_amd import io_native

FPAIRED, FMUNMAP, FREAD1, FREAD2 = 0x1, 0x8, 0x40, 0x80


def join_tasks(desc, d_first, task, h_flags):
    """-> (jtask, drop) of the device's descriptors: d_first [n_sub + 1] per walk task, task [n_sub, 10] the walk plan, h_flags [n_host]"""
    n = int(d_first[-1])
    jt = np.zeros(n, np.int64)
    drop = np.zeros(n, bool)
    for u in range(task.shape[0]):
        a, b = int(d_first[u]), int(d_first[u + 1])
        host = int(task[u, 9])
        jt[a:b] = host
        if h_flags[host]:
            drop[a:b] = True
        elif u > 0 and int(task[u - 1, 9]) == host:
            drop[a:b] = desc["pos"][a:b] < int(task[u - 1, 1])  # (the sub-task before met it too, and kept it)
    return jt, drop


class Reach:
    """the reach intervals of the first walk, flat and sorted by (reference, start), each with its stage task"""

    def __init__(self, task, reach):
        n = reach.shape[0]
        self.a, self.b = reach[:, 0].astype(np.int64), reach[:, 1].astype(np.int64)
        self.tid, self.host = np.full(n, -1, np.int64), np.full(n, -1, np.int64)
        for u in range(task.shape[0]):
            self.tid[task[u, 4]: task[u, 5]] = task[u, 0]
            self.host[task[u, 4]: task[u, 5]] = task[u, 9]
        key = self.tid * (1 << 32) + self.b
        assert (np.diff(key) > 0).all(), "reach intervals must ascend by (reference, end)"
        self.key = key

    def covering(self, tid, pos):
        i = int(np.searchsorted(self.key, tid * (1 << 32) + pos, side="right"))  # first interval of (tid) with b > pos
        if i < self.key.size and self.tid[i] == tid and self.a[i] <= pos:
            return int(self.host[i])
        return -1


def run(lib, stage, desc, d_first, d_flags, plan, n_ref, all_bases=False):
    """-> dict(voff, qname, mate, bases, n_qnames, src, lookups) of the kept records in output order"""
    task, reach = plan["task"], plan["reach"]
    n_host = int(task[:, 9].max()) + 1 if task.shape[0] else 0
    h_flags = np.zeros(max(1, n_host), np.int32)
    tot = (C.c_int64 * 2)()
    d_flags = np.ascontiguousarray(d_flags, np.int32)
    io_native._check(lib, lib.uz_stage_walk_flagged(stage, d_flags.ctypes.data, h_flags.ctypes.data, tot))

    def extras(d0, a0):
        x = np.zeros(max(1, int(tot[0]) - d0), io_native.WALK_DESC)
        io_native._check(lib, lib.uz_stage_extra(stage, d0, a0, x.ctypes.data, None, None))
        return x[: int(tot[0]) - d0]

    jt, drop = join_tasks(desc, d_first, task, h_flags)
    x = extras(0, 0)
    D = np.concatenate([desc[: int(d_first[-1])], x])
    jt = np.concatenate([jt, (x["task"] & 0x7FFFFFFF).astype(np.int64)])
    drop = np.concatenate([drop, np.zeros(x.size, bool)])
    assert (x["task"] & io_native.WALK_TASK_JOIN).all()
    R = Reach(task, reach)
    keep = np.where(drop, 0, np.where(D["direct"] != 0, 2, 0)).astype(np.int8)
    mate = np.full(D.size, -2, np.int64)
    target = {}  # member -> the look-up task that answers it

    def groups_of(D, drop):
        g = {}
        for i in np.flatnonzero(~drop):
            g.setdefault(int(D["h1"][i]), []).append(int(i))
        return g

    groups = groups_of(D, drop)
    frontier = [int(i) for i in np.flatnonzero(keep == 2)]
    n_lookups = 0
    for _round in range(64):
        need = []
        while frontier:
            nxt = []
            for i in frontier:
                if mate[i] != -2:
                    continue
                xr = D[i]
                fl = int(xr["flag"])
                # a copy that will be folded away -- another copy of the same record is a member and wins the fold -- asks for nothing (without this
                # rule two mates outside every reach interval look each other up for ever: every answer of the index is a new copy)
                folds = any(e != i and D["voff"][e] == xr["voff"] and (keep[e] > keep[i] or (keep[e] == keep[i] and e < i)) for e in groups.get(int(xr["h1"]), ()))
                if folds or not ((fl & FPAIRED) and not (fl & FMUNMAP) and 0 <= int(xr["mtid"]) < n_ref):
                    mate[i] = -1
                    continue
                tc = target[i] if i in target else R.covering(int(xr["mtid"]), int(xr["mpos"]))
                if tc < 0:
                    need.append(i)
                    continue
                want = (fl ^ (FREAD1 | FREAD2)) & (FREAD1 | FREAD2)
                seen, best = False, -1
                for e in groups.get(int(xr["h1"]), ()):
                    y = D[e]
                    if jt[e] != tc or y["h2"] != xr["h2"] or y["l_name"] != xr["l_name"]:
                        continue
                    seen = True
                    if int(y["pos"]) < int(xr["mpos"]) + 1 and int(y["end"]) > int(xr["mpos"]) and (int(y["flag"]) & want):
                        if best < 0 or y["voff"] < D["voff"][best]:
                            best = e
                if not seen and i not in target:
                    need.append(i)
                    continue
                mate[i] = best
                if best >= 0 and keep[best] == 0:
                    keep[best] = 1
                    nxt.append(best)
            frontier = nxt
        if not need:
            break
        nr = np.zeros(len(need), io_native.NEED_REC)
        for k, i in enumerate(need):
            nr[k] = (D["h1"][i], D["h2"][i], D["l_name"][i], D["mtid"][i], D["mpos"][i], i, 0)
        ans = np.zeros(len(need), np.int32)
        d0 = int(tot[0])
        io_native._check(lib, lib.uz_stage_lookup(stage, len(need), nr.ctypes.data, ans.ctypes.data, tot))
        n_lookups += len(need)
        x = extras(d0, 0)
        D = np.concatenate([D, x])
        jt = np.concatenate([jt, (x["task"] & 0x7FFFFFFF).astype(np.int64)])
        drop = np.concatenate([drop, np.zeros(x.size, bool)])
        keep = np.concatenate([keep, np.zeros(x.size, np.int8)])
        mate = np.concatenate([mate, np.full(x.size, -2, np.int64)])
        groups = groups_of(D, drop)
        for k, i in enumerate(need):
            target[i] = int(ans[k])
        frontier = list(need)
    # ---- the kept records: one per virtual offset, the copy a fetch returned first
    kept = np.flatnonzero(keep != 0)
    order = kept[np.lexsort((kept, -keep[kept].astype(np.int64), D["voff"][kept]))]  # by voff, then the higher keep, then the earlier copy
    v = D["voff"][order]
    first = np.ones(order.size, bool)
    first[1:] = v[1:] != v[:-1]
    surv = order[first]
    gidx = np.full(D.size, -1, np.int64)
    run_id = np.cumsum(first) - 1
    gidx[order] = run_id  # (a folded copy stands for its survivor)
    # ---- names by first appearance
    K = surv.size
    first_of = np.zeros(K, np.int64)
    for k, i in enumerate(surv):
        best = k
        for e in groups[int(D["h1"][i])]:
            if gidx[e] >= 0 and D["h2"][e] == D["h2"][i] and D["l_name"][e] == D["l_name"][i]:
                best = min(best, int(gidx[e]))
        first_of[k] = best
    is_first = first_of == np.arange(K)
    ids = np.cumsum(is_first) - is_first
    qname = ids[first_of]
    m = mate[surv]
    mate_out = np.where(m >= 0, gidx[np.maximum(m, 0)], -1)
    bases = (keep[surv] == 2) | bool(all_bases)
    return dict(voff=D["voff"][surv], qname=qname.astype(np.uint32), mate=mate_out.astype(np.int32), bases=bases, n_qnames=int(is_first.sum()),
                src=D["src"][surv], lookups=n_lookups, n_extra=int(tot[0]), h_flags=h_flags, trips=_round)
