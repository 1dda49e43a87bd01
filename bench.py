#!/usr/bin/env python
"""bench.py -- phased DNMs/s of the per-DNM phasing path on MI355X (BASELINE.json metric).

Metric 1: input DNMs / wall time of { H2D of the pre-decoded SoA, K1-K5, D2H, decision }.
One "step" is one whole pass of the hot path over one batch of synthetic DNMs, and `value` is measured with the
staging INSIDE the timed region:
    sites + genotype columns: pinned host -> HBM, K1 site scan, K2 window emit, het lists back to the host
    (they tell the decoder which fetches the read stage makes), then per chunk of DNMs: the alignment records those
    fetches can return, in the staged (packed) form, pinned host -> HBM on the copy stream while the previous
    chunk's K3a + per-DNM kernel run on the compute stream; per-DNM results back to the host.
What is NOT timed is decode: producing the pinned columns (here: the synthetic generator + the host-side
fetch-reach selection of libunfazed_io), reported as `decode_s`.  `value_resident` is the same pass with every
input already in HBM (kernel-pass throughput; the round-1 headline).

Workload (BASELINE.json configs[2]): N synthetic SNV/INDEL DNMs (90/10) of one trio placed
UNIFORMLY over a whole-genome-like sites table (default 20 M sites, 24 contigs) -- about a third of the +-6 kb read
windows overlap a neighbour's and share its records -- 30x paired-end pile-up, extended read-backed phasing on.

Multi-GPU: DNMs shard embarrassingly, one process per GPU, no collective on the data path; the only
communication is the barrier and the max-over-ranks of the elapsed time.  `--scaling weak` (default): every rank
phases its own N DNMs; `--scaling strong` (config 4): the SAME N DNMs cut into contiguous shards.
`python bench.py --gpus N` without a launcher starts the N ranks itself.

Extra objects on the JSON line: `roofline` for the K1 site-scan kernel (HBM-bound; algorithmic bytes = 20 B/site),
`issue_model` for the read stage, `link` (achieved host-link GB/s of the staged pass), and `cpu_baseline`: the CPU
oracle (a C port of the reference's algorithm, oracle/) timed on this box's host cores on a bounded sample of the
same DNMs, whose results are also compared with the GPU's (parity at bench scale).
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # (the second step of a run still allocates: lists of three finds, table blocks; measured 1 / 2 / 3 warm-up steps: 16.7 / 15.3 / 15.4 ms)
    ap.add_argument("--dnms", type=int, default=int(os.environ.get("UZ_BENCH_DNMS", 100000)),
                    help="DNMs per GPU (weak scaling) / in total (strong scaling)")
    ap.add_argument("--sites", type=int, default=int(os.environ.get("UZ_BENCH_SITES", 20000000)))
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong when --gpus > 1 (BASELINE configs[3]: the SAME 100 k DNMs cut into contiguous shards), weak = every rank its own --dnms")
    ap.add_argument("--workload", choices=["snv", "cnv"], default="snv",
                    help="snv: BASELINE configs[2] (100k SNV/INDEL DNMs, read-backed); cnv: configs[4] (10k DEL/DUP: allele-balance "
                         "K6 + the SV read-backed stage around both breakpoints)")
    ap.add_argument("--chunks", type=int, default=0, help="DNM chunks of the staged pass (uploads overlap the kernels); 0 = from the shard size "
                    "(shard.chunk_plan: >= 12.5 k DNMs per chunk, at least 3)")
    ap.add_argument("--sites16", action="store_true", help="staged pass: the genotype columns of the site windows in 16 bits (default: the eight-bit link form)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the streamed form of the staged pass (`streamed` in the line)")
    ap.add_argument("--first-chunk", type=float, default=None, help="size of the first chunk of the staged pass relative to the others (default: shard.chunk_plan's: 1.0 for a batch of four chunks or more, else 0.5)")
    ap.add_argument("--last-chunk", type=float, default=0.7, help="size of the last chunk of the staged pass relative to the others")
    ap.add_argument("--one-site-table", action="store_true", help="staged pass: one site stage for the whole batch in front of the chunks (default: a site stage per chunk, pipelined with the record uploads)")
    ap.add_argument("--cpu-dnms", type=int, default=60000, help="DNMs in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-staged", action="store_true", help="resident pass only (profiling runs)")
    ap.add_argument("--no-config5", action="store_true", help="default snv run: do not also run BASELINE configs[4] (10 k DEL/DUP) as a child process for the `config5` object")
    ap.add_argument("--feed-dnms", type=int, default=int(os.environ.get("UZ_BENCH_FEED_DNMS", 20000)),
                    help="DNMs of the files -> results pass (`feed` / `value_e2e`): their pile-ups are written as a real BAM + BAI, the sites table as "
                         "a BGZF VCF + TBI, and decoded back through the indexes inside the timed region (0 = skip)")
    ap.add_argument("--feed-filler-dnms", type=int, default=int(os.environ.get("UZ_BENCH_FEED_FILLER_DNMS", 5000)),
                    help="DNMs of the `feed_filler` pass: the files -> results pass on a BAM whose gaps between the pile-ups are filled with read pairs (0 = skip)")
    ap.add_argument("--feed-filler", type=float, default=30.0, help="coverage of the filler between the pile-ups of the `feed_filler` pass")
    ap.add_argument("--feed-chunk", type=int, default=3400, help="DNMs per chunk of the files -> results pass (scripts/feed_sweep.sh, two boxes: 1500 / 2500 / 3400 / 4000 / 5000 = 23.7 / 24.6 / 25.9 / 24.3 / 22.5 k DNMs/s)")
    ap.add_argument("--feed-reps", type=int, default=3, help="timed passes of the feed leg (the median is reported)")
    ap.add_argument("--feed-walk", choices=("device", "host"), default="device",
                    help="feed pass: where the BAM records are walked -- device: the inflated blocks stay in HBM, k_bam_walk hands the host 64-byte descriptors, "
                         "the table is unpacked from HBM (include/uz_bamwalk.h); host: uz_bam_stage_* walks them on the host's cores (the link form)")
    ap.add_argument("--feed-joins", choices=("device", "host"), default="device",
                    help="feed pass, device walk: where mate() + closure + name numbering run -- device: over the descriptors where they lie in HBM (csrc/k_bamjoin.hip: "
                         "nothing of the records crosses the link); host: the descriptors come down, uz_bam_stage_finish_sub joins them, the kept list goes up (round 5)")
    ap.add_argument("--feed-inflate", choices=("device", "host"), default="device",
                    help="feed pass: who inflates the BGZF blocks of the BAM -- the device (uz_bgzf_inflate_to_host) or the host's cores")
    ap.add_argument("--feed-level", type=int, default=6, help="deflate level of the files written for the feed pass (samtools / bgzip default: 6)")
    ap.add_argument("--feed-dir", default=None, help="where the files of the feed pass go (default: a temporary directory in /dev/shm, else /tmp)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks as children BEFORE anything touches the GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll all of them: when one rank dies the others would sit in the barrier forever -- end them
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = max(rc, abs(r))
                for q in live:
                    q.terminate()
    sys.exit(rc)


def bind_near_gpu(torch, local_rank):
    """Run (and allocate the pinned staging memory: first touch) on the CPUs of the GPU's own NUMA node: a host-to-device copy
    out of the other socket's memory crosses the socket link first.  Best effort (sysfs); UZ_BENCH_NO_NUMA=1 leaves the process
    where the launcher put it.  -> the node, or None"""
    if os.environ.get("UZ_BENCH_NO_NUMA"):
        return None
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        # ... and, for the feed pass under a CPU quota (a container granted q CPUs of a much larger machine), a COMPACT set of them: its host side is
        # a chain of multi-threaded sections that hand each other gigabytes (descriptors -> records -> name ids -> kept list); 2 q threads scattered
        # over every core complex of the socket share no L3 (measured, 16-CPU quota on a 128-CPU node: 64 k DNMs/s from files anywhere on the node,
        # 73 k on its first 32 CPUs).  UZ_BENCH_NO_COMPACT=1 keeps the whole node.
        try:
            from unfazed_amd import io_native
            q = int(io_native.cpu_quota() or 0)
        except Exception:
            q = 0
        bind_near_gpu.compact = None
        if q > 0 and len(cpus) > 2 * q and not os.environ.get("UZ_BENCH_NO_COMPACT"):
            bind_near_gpu.compact = set(sorted(cpus)[: 2 * q])  # (the feed pass runs there: its host side is what bounds it)
        bind_near_gpu.before = os.sched_getaffinity(0)
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def main():
    args = parse()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.workload == "cnv" and not args.chunks:
        if args.first_chunk is None:
            args.first_chunk = 0.5
        args.chunks = 2  # (rounds 3 / 4: three -- 2 / 3 / 4 / 6 chunks = 6.6 / 6.2 / 7.4 / 8.1 ms; the read stage of an SV chunk is as long as its slowest
        # event -- a breakpoint pile-up of a thousand records in one wave -- so small chunks cost more in total.  End of round 5, with the find's answers
        # past the DMA queue: 2 equal chunks / 2 with the second 1.5 x / 3 with a half-size first = 4.29 / 4.31 / 4.45 ms)
    if args.workload == "cnv" and "--last-chunk" not in " ".join(sys.argv):
        # EQUAL chunks: the read stage of an SV chunk is as long as its slowest event, so a smaller last chunk buys nothing and makes
        # the others larger (scripts/cnv_sweep.sh, header build on its own stream: last chunk 0.3 / 0.5 / 0.7 / 1.0 x = 5.44 / 5.25 / 5.06 /
        # 4.95 ms; with that, a first chunk of 0.5 / 0.7 / 1.0 x = 5.03 / 4.90 / 4.93 ms; 2 / 4 equal chunks 4.97 / 5.19 ms)
        args.last_chunk = 1.0
    if args.workload == "cnv" and "--dnms" not in " ".join(sys.argv) and "UZ_BENCH_DNMS" not in os.environ:
        args.dnms = 10000
    world = int(os.environ.get("WORLD_SIZE", 0))
    if world == 0:
        if args.gpus > 1:
            spawn_ranks(args)
        world = 1
    elif world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    # development aid for a box with ONE GPU: UZ_BENCH_ONE_DEVICE=1 puts every rank on device 0 and runs the barrier / max-over-ranks over gloo (two
    # ranks cannot share a device under RCCL) -- it exercises the N-rank code path (shards, barrier, reduction, rank 0's line), it measures nothing
    one_device = bool(os.environ.get("UZ_BENCH_ONE_DEVICE"))
    if one_device:
        local_rank = 0
    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X GPU: there is no CPU fallback for the phasing path")
    torch.cuda.set_device(local_rank)
    numa = bind_near_gpu(torch, local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from synth.benchload import BenchLoad
    from unfazed_amd import abi, build, io_native, pipeline, shard
    from unfazed_amd.engine import (HipEngine, K_PHASE, K_SITE_SCAN, K_SIZING, K_WINDOW_COUNT, K_WINDOW_FILL, K_CNV, PinnedPool)

    build.build()
    build.build_io()
    t_gen = time.time()
    cnv = args.workload == "cnv"
    seed_off = 0 if args.scaling == "strong" else 1000 * rank
    lo = hi = None
    if args.scaling == "strong":
        # config 4: the same list on every rank, cut into contiguous shards (sites replicated per GPU)
        b = shard.shard_bounds(args.dnms, world)
        lo, hi = b[rank], b[rank + 1]
    load = BenchLoad(args.dnms, args.sites, workload=args.workload, seed_off=seed_off, lo=lo, hi=hi, device=local_rank)
    sc, ev, dn, cl, cfg, wl, per_ev = load.sc, load.ev, load.dn, load.cl, load.cfg, load.wl, load.per_ev
    t_gen = time.time() - t_gen

    eng = HipEngine(local_rank)
    P = abi.make_params()  # the reference's CLI defaults
    eng.set_params(P)
    sid, fid, rid = load.adopt(eng, P)
    cutoff = load.cutoff
    n = ev.n
    mode = abi.FIND_SECOND_WINDOW
    ev_vt, ev_refs, ev_alts = load.ev_vt, load.ev_refs, load.ev_alts
    dv = load.view_of(0, n)

    def with_cnv(f, r):
        """config 5: K6 over the batch, merged with the read-backed counts as summarize_record merges them"""
        if not cnv:
            return r
        k = eng.phase_cnv(f, dv, P, rb_counts=r["counts"], want_lists=False)
        return dict(status=r["status"], counts=r["counts"], origin=k["origin"], evidence=k["evidence"], etype=k["etype"],
                    cnv_counts=k["cnv_counts"])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    def timed(step):
        for _ in range(args.warmup):
            res = step()
        # Inside the timed region only the roofline kernel (K1) carries HIP events; the per-kernel table of the line comes from a
        # few more steps behind it with every id timed (two event records per launch are host work in front of ~40 launches per
        # chunk -- on the small chunks of config 5 the host's queueing IS on the critical path)
        eng.prof_enable([K_SITE_SCAN])
        eng.prof_reset()
        # (the cyclic collector stays out of the timed region, as it stays out of the product's phasing calls -- session.no_gc_pauses: a full
        # collection walks every container this process holds, the generator's tables' wrappers included, and lands in whichever step crosses
        # the allocation threshold)
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        barrier()
        t0 = time.perf_counter()
        c0 = time.process_time()
        for _ in range(args.steps):
            res = step()
        barrier()
        elapsed = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        timed.host_cpu_s = (time.process_time() - c0) / max(1, args.steps)  # CPU seconds of this rank's process per step, all its threads
        timed.per_rank = [elapsed]
        if dist is not None:
            t = torch.zeros(world, dtype=torch.float64, device="cpu" if one_device else "cuda")
            t[rank] = elapsed
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            timed.per_rank = [float(x) for x in t.tolist()]
            elapsed = max(timed.per_rank)
        k1 = eng.prof_get(K_SITE_SCAN)
        eng.prof_enable(True)
        eng.prof_reset()
        extra = max(1, min(3, args.steps))
        for _ in range(extra):
            step()
        eng.sync()
        # (totals scaled to the timed region's step count: the readers below divide by args.steps)
        prof = {k: tuple(x * args.steps / extra for x in eng.prof_get(k)) for k in (K_WINDOW_COUNT, K_WINDOW_FILL, K_PHASE, K_SIZING, K_CNV)}
        prof[K_SITE_SCAN] = k1
        timed.hbm_build_dnms = int(eng.prof_units(K_PHASE))
        eng.prof_enable(False)
        return res, elapsed, prof

    # ---------------------------------------------------------------- resident pass
    rtrace = [] if os.environ.get("UZ_BENCH_TRACE") else None

    def step_resident():
        if rtrace is None:
            eng.drop_derived()
            return with_cnv(fid, eng.phase_raw(fid, rid, dv, P, mode))
        t0 = time.perf_counter()
        eng.drop_derived()
        t1 = time.perf_counter()
        r = eng.phase_raw(fid, rid, dv, P, mode)
        t2 = time.perf_counter()
        r = with_cnv(fid, r)
        rtrace.append([round((b - a) * 1e3, 2) for a, b in ((t0, t1), (t1, t2), (t2, time.perf_counter()))])
        return r

    res_r, el_r, prof_r = timed(step_resident)
    host_cpu_resident = timed.host_cpu_s
    if rtrace:
        print("[resident step, ms] drop_derived | read stage | allele balance:", rtrace, file=sys.stderr)
    per_rank_r = list(timed.per_rank)

    # ---------------------------------------------------------------- staged pass
    # (the pass itself is the product's: unfazed_amd/pipeline.py; what the decoders would leave in pinned memory -- the site windows and the
    # records of every chunk in the link form -- comes from synth/benchload.py, outside the timing: `decode_s`)
    staged = None
    if not args.no_staged:
        t_dec = time.time()
        pool = PinnedPool()
        chunks, st = load.stage(eng, P, mode, fid, pool, chunks=args.chunks, last_chunk=args.last_chunk, first_chunk=args.first_chunk,
                                sites16=args.sites16, per_chunk_sites=not args.one_site_table, log_first=bool(os.environ.get("UZ_BENCH_LINK_BYTES")))
        f_all = s_all = None
        if args.one_site_table:  # one site stage for the whole batch in front of the chunks (a measurement aid: the default pipelines it per chunk)
            held, hs, hg, wide, ns = load.pinned_sites(pool, load.window_sites(0, n, P), eight_bit=not args.sites16)
            st["sites"], st["site_bytes"] = ns, ns * (4 + 1 + 1 + 1 + 1 + (18 if args.sites16 else 9))
        t_dec = time.time() - t_dec
        trace = [] if os.environ.get("UZ_BENCH_TRACE") else None

        def step_staged():
            if not args.one_site_table:
                return pipeline.run_pipelined(eng, P, mode, n, chunks, cnv=cnv, trace=trace)
            s2, f2 = eng.upload_sites_family_async(held, hs["gt"], hg["rd"], hg["ad"], hg["gq"], wide)
            out = pipeline.run_pipelined(eng, P, mode, n, chunks, cnv=cnv, fid=f2, trace=trace)
            eng.free_sites(s2)
            return out

        res_s, el_s, prof_s = timed(step_staged)
        per_rank_s = list(timed.per_rank)
        host_cpu_staged = timed.host_cpu_s
        if trace:
            print("[staged step, ms] site stage 0, then per chunk: find (behind the read stage of chunk k - 2) | its results | enqueue records, the site windows of chunk k + 2 | queue the read stage of chunk k - 1 (config 5: + allele balance of chunk k - 2); last read stages:", trace[-2:], file=sys.stderr)
        mism = sum(int((np.asarray(res_s[k]) != np.asarray(res_r[k])).sum()) for k in res_r)
        # The same steps as ONE stream of chunks (what a cohort run hands the pipeline: batch after batch): the first uploads of a step travel beside
        # the last read stages of the step before, so the head and the tail of the pipeline are paid once, not per step.  Reported beside `value`
        # (which stays the step-by-step pass: every step collected before the next one starts), never instead of it.
        streamed = None
        if not args.one_site_table and not args.no_streamed and world == 1:
            S = max(2, args.steps)
            rep = [dict(c, a=c["a"] + s_ * n, b=c["b"] + s_ * n) for s_ in range(S) for c in chunks]
            pipeline.run_pipelined(eng, P, mode, 2 * n, rep[: 2 * len(chunks)], cnv=cnv)  # (warm-up: two steps' worth)
            eng.sync()
            import gc
            gc.collect()
            gc_was = gc.isenabled()
            gc.disable()
            t0 = time.perf_counter()
            out_st = pipeline.run_pipelined(eng, P, mode, S * n, rep, cnv=cnv)
            eng.sync()
            el_st = time.perf_counter() - t0
            if gc_was:
                gc.enable()
            mism_st = sum(int((np.asarray(out_st[k][s_ * n:(s_ + 1) * n]) != np.asarray(res_r[k])).sum()) for k in res_r for s_ in range(S))
            streamed = {"value": round(n * S / el_st, 1), "unit": "DNMs/s", "ms_per_step": round(el_st / S * 1e3, 3), "steps": S, "chunks": len(rep),
                        "result_mismatches_vs_resident": int(mism_st),
                        "note": "the same %d steps handed to the pipeline as one stream of chunks (batch after batch, as a cohort run does): a step's first uploads overlap the read stages of the step before; `value` is the step-by-step pass" % S}
        staged = dict(streamed=streamed, host_cpu_s=host_cpu_staged, elapsed=el_s, prof=prof_s, bytes=st["read_bytes"] + st["site_bytes"], read_bytes=st["read_bytes"], records=st["records"],
                      decode_s=t_dec, mismatches_vs_resident=mism, chunks=len(chunks),
                      sites=st["sites"], site_stage="whole batch" if args.one_site_table else "per chunk")
        res = res_s
    else:
        res = res_r

    n_total = n * world if args.scaling == "weak" else sum(np.diff(shard.shard_bounds(args.dnms, world)))
    value_resident = n_total * args.steps / el_r
    ms_resident = el_r / args.steps * 1e3
    if staged:
        value = n_total * args.steps / staged["elapsed"]
        ms_per_step = staged["elapsed"] / args.steps * 1e3
    else:
        value, ms_per_step = value_resident, ms_resident
    status = res["status"]
    called = (res["origin"] == abi.OR_DAD) | (res["origin"] == abi.OR_MOM)
    called &= ((res["etype"] & abi.ET_AMBIG_FLAG) == 0) if cnv else (status == abi.ST_OK)
    phased = int(called.sum())
    truth = np.where(ev.origin == 0, abi.OR_DAD, abi.OR_MOM)
    correct = int((res["origin"][called] == truth[called]).sum())

    def kern_ms(prof):
        return {"site_scan": round(prof[K_SITE_SCAN][0] / args.steps, 3),
                "window_count+scan": round(prof[K_WINDOW_COUNT][0] / args.steps, 3),
                "window_fill": round(prof[K_WINDOW_FILL][0] / args.steps, 3),
                "sizing": round(prof[K_SIZING][0] / args.steps, 3),
                "phase": round(prof[K_PHASE][0] / args.steps, 3),
                "cnv_count": round(prof[K_CNV][0] / args.steps, 3)}

    k1_ms, k1_n = prof_r[K_SITE_SCAN]
    k1_us = k1_ms / max(1, k1_n) * 1e3
    bytes_per_site = 20.0  # 19 B read (packed trio GT + 9 x u16) + 1 B class written; DESIGN.md "K1"
    achieved = bytes_per_site * sc.n / (k1_us * 1e-6) / 1e9 if k1_n else 0.0
    # counters come from a committed profile (separate --pmc passes, scripts/profile_round.sh) and are quoted only while that profile
    # measured THIS build's kernels (it records the hash of the device sources); otherwise null + where to look
    ksha = build.kernel_source_hash()
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if int(tj.get("n_sites", -1)) == sc.n and tj.get("kernel_source_sha") == ksha:
                traffic = tj.get("hbm_bytes_per_launch")
            else:
                traffic_note = "profiles/k1_traffic.json was collected on other kernel sources (%s) or another table size: not quoted" % tj.get("kernel_source_sha")
        except Exception:
            traffic = None
    # two byte counts: `achieved` / `frac` on the bytes the kernel moves (20 B per site: it does not read `pos`), and the same launch priced at
    # SURVEY.md 8(d)'s fixed 24 B per site (pos + packed GT + 9 x u16 + class)
    survey_bytes = 24.0 * sc.n
    roofline = {"bound": "hbm", "kernel": "k_site_scan", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(achieved / 8000.0, 4), "traffic": traffic, "avg_launch_us": round(k1_us, 2),
                "algorithmic_bytes_per_launch": int(bytes_per_site * sc.n), "bytes_moved": int(bytes_per_site * sc.n),
                "algorithmic_bytes_survey": int(survey_bytes),
                "achieved_on_survey_bytes": round(survey_bytes / (k1_us * 1e-6) / 1e9, 1) if k1_n else 0.0,
                "frac_on_survey_bytes": round(survey_bytes / (k1_us * 1e-6) / 1e9 / 8000.0, 4) if k1_n else 0.0,
                "kernel_source_sha": ksha}
    if traffic_note:
        roofline["traffic_note"] = traffic_note

    # k_phase has no GB/s figure: it is bound by instruction issue (integer VALU work) with memory latency on top.  The
    # counters of the same workload (profiles/phase_issue.json, scripts/profile_round.sh) give the floor: VALU
    # wave-instructions x 4 cycles / 1024 SIMDs at the measured shader clock; frac = floor / this run's kernel time.
    issue_model = None
    tpath = os.path.join(ROOT, "profiles", "phase_issue.json")
    ph_ms, ph_n = prof_r[K_PHASE]
    if os.path.exists(tpath) and ph_n:
        try:
            tj = json.load(open(tpath))
            if int(tj.get("dnms", -1)) == n and tj.get("kernel_source_sha") != ksha:
                issue_model = {"kernel": "k_phase", "stale": True, "source": "profiles/phase_issue.json",
                               "note": "collected on other kernel sources (%s): not quoted for this build (%s)" % (tj.get("kernel_source_sha"), ksha)}
            elif int(tj.get("dnms", -1)) == n:
                measured = ph_ms / ph_n
                issue_model = {"kernel": "k_phase", "bound": "instruction issue (one wavefront per DNM, as many DNMs per SIMD as the LDS arenas leave room for: waves_per_simd); measured_ms covers the three launches of a batch (arena build, its larger-arena rerun, HBM build)",
                               "valu_wave_instructions_per_dnm": round(tj["valu_wave_instructions_per_dnm_with_candidates"], 0),
                               "valu_pipe_busy_frac_profiled": round(tj["valu_pipe_busy_frac"], 4),
                               "valu_lane_utilisation": round(tj["valu_lane_utilisation"], 4),
                               "wave_cycles_parked_frac": round(tj["wave_cycles_parked_frac"], 4),
                               "waves_per_simd": round(tj["waves_per_simd"], 2), "shader_clock_GHz": round(tj["shader_clock_GHz"], 3),
                               "valu_floor_ms": round(tj["valu_floor_ms"], 3), "measured_ms": round(measured, 3),
                               "frac": round(tj["valu_floor_ms"] / measured, 4), "source": "profiles/phase_issue.json"}
        except Exception:
            issue_model = None

    feed = feed_filler = None
    if rank == 0 and world == 1 and not cnv and args.feed_dnms > 0:
        if getattr(bind_near_gpu, "compact", None):  # the files -> results pass is host work: 2 q CPUs next to each other, next to the GPU
            os.sched_setaffinity(0, bind_near_gpu.compact)
        elif getattr(bind_near_gpu, "before", None):  # ... or every core of the box
            os.sched_setaffinity(0, bind_near_gpu.before)
        feed = feed_e2e(args, eng, sc, ev, cl, cfg, P, mode, res_r, PinnedPool)
        if feed is not None:
            feed["host_cpus_bound_to"] = len(os.sched_getaffinity(0))
        if feed is not None and args.feed_filler_dnms > 0 and args.feed_walk == "device" and args.feed_inflate == "device":
            # The same pass on a file that looks like a file: the first --feed-filler-dnms DNMs, once with the gaps between their
            # pile-ups filled at 30 x -- bins with a lead-in, records to walk past between two reach intervals -- and once without, so that the two
            # differ in nothing but the filler
            ff = feed_e2e(args, eng, sc, ev, cl, cfg, P, mode, res_r, PinnedPool, dnms=args.feed_filler_dnms, filler=args.feed_filler, lite=True, cutoff_of_the_pile_ups=cutoff)
            f0 = feed_e2e(args, eng, sc, ev, cl, cfg, P, mode, res_r, PinnedPool, dnms=args.feed_filler_dnms, filler=0.0, lite=True)
            ff["same_dnms_without_filler"] = {k: f0[k] for k in ("value_e2e", "seconds_of_every_pass", "result_mismatches_vs_resident", "blocks_inflated_per_dnm",
                                                               "inflated_MB_per_dnm", "records_walked_per_record_kept", "link_bytes_per_dnm", "host_cpu_seconds_per_pass")}
            ff["same_dnms_without_filler"]["bam"] = f0["bam"]
            ff["note"] = ("read pairs at %g x in every gap between the pile-ups (and 64 kb beyond a contig's first / last one): what bamfile.fetch meets on a whole-genome "
                          "file, read_collector.py:385, :167 -- the walk starts where the linear index puts the first record of a 16 kb window and walks past "
                          "what lies in front of the reach" % args.feed_filler)
            feed_filler = ff

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and args.cpu_dnms > 0:
        if getattr(bind_near_gpu, "before", None):  # the CPU baseline gets every core of the box back
            os.sched_setaffinity(0, bind_near_gpu.before)
        cpu = cpu_baseline(args, wl, sc, ev, dn, per_ev, cl, cfg, P, cutoff, res, ev_vt, ev_refs, ev_alts, cnv)
        if cpu and feed and feed.get("cpu_decode_s") is not None and cpu.get("value"):
            # the CPU path measured the same way: the same files decoded on the host (every thread of the box), then the oracle at
            # its best thread count -- back to back, not overlapped (the port holds its tables in memory whole)
            m_f = feed["dnms"]
            t_cpu = feed["cpu_decode_s"] + m_f / cpu["value"]
            cpu["value_e2e"] = round(m_f / t_cpu, 1)
            cpu["e2e"] = ("files -> results for the %d DNMs of the feed pass: %.2f s decode (uz_bam_decode of the same BAM, %d records, + uz_vcf_decode_regions of "
                          "the same windows, all host threads) + %.2f s oracle at its best thread count" % (m_f, feed["cpu_decode_s"], feed["cpu_decode_records"], m_f / cpu["value"]))

    config5 = None
    if rank == 0 and world == 1 and not cnv and not args.no_config5 and args.dnms == 100000:
        # BASELINE configs[4] in the same driver-run line: the 10 k DEL/DUP workload as a child process (its own generator state and
        # contexts; this process has let go of nothing it needs and waits)
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "cnv", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-config5",
                   "--cpu-dnms", "10000" if not args.no_cpu else "0"] + (["--no-cpu"] if args.no_cpu else [])
            cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
            line5 = [x for x in cp.stdout.decode().splitlines() if x.startswith("{")]
            j5 = json.loads(line5[-1]) if line5 else None
            if j5:
                config5 = {k: j5.get(k) for k in ("value", "unit", "ms_per_step", "value_resident", "ms_per_step_resident", "kernels_ms_per_step", "calls", "cpu_baseline")}
                config5["workload"] = j5["config"]["workload"]
                config5["link"] = j5.get("link")
                config5["metric"] = "phased events/sec (DEL/DUP: allele balance K6 merged with the SV read-backed stage)"
        except Exception as e:  # the headline line must not die with the side run
            config5 = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "phased DNMs/sec", "value": round(value, 1), "unit": "DNMs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "n/a", "vs_baseline": None, "dtype": "u8/int32 (+f64 allele balance)",
            "data": "synthetic",
            "config": {"workload": (("10k CNV (DEL/DUP) DNMs exercising the allele-balance sv_phaser path (BASELINE configs[4]): K6 + SV read-backed stage"
                                     if args.dnms == 10000 else "%d synthetic DEL/DUP events, allele balance + SV read-backed stage" % args.dnms) if cnv else
                                    ("100k synthetic SNV/INDEL DNMs, whole-genome sites VCF, --no-extended off (BASELINE configs[2])"
                                     if args.dnms == 100000 else "%d synthetic SNV/INDEL DNMs, extended on" % args.dnms))
                       + ("; staged: H2D of the pre-decoded columns inside the timed region" if staged else "; inputs resident in HBM"),
                       "dnms_per_gpu": n, "sites": sc.n, "coverage": "30x", "search_dist": 5000, "dnm_placement": "uniform",
                       "read_clusters": cl.n, "dnms_sharing_a_cluster": int(cl.nd[cl.nd > 1].sum()),
                       "alignment_records": wl.n_segs, "parallelism": "dnm-shard x%d, no collective" % world},
            "value_resident": round(value_resident, 1), "ms_per_step_resident": round(ms_resident, 3),
            # CPU seconds of rank 0's process per step (every thread of it): what eight ranks on one node would ask of the host's cores between them
            "host_cpu_seconds_per_step": {"staged": round(staged["host_cpu_s"], 5) if staged else None, "resident": round(host_cpu_resident, 5)},
            "ms_per_step_by_rank": [round(x / args.steps * 1e3, 3) for x in (per_rank_s if staged else per_rank_r)],
            "ms_per_step_resident_by_rank": [round(x / args.steps * 1e3, 3) for x in per_rank_r],
            "roofline": roofline,
            "issue_model": issue_model,
            "cpu_baseline": cpu,
            "value_e2e": feed["value_e2e"] if feed else None,
            "feed": feed,
            "feed_filler": feed_filler,
            "config5": config5,
            "kernels_ms_per_step": kern_ms(prof_r),
            "calls": {"phased": phased, "correct_vs_truth": correct, "status_counts": np.bincount(status, minlength=6).tolist(),
                      "dnms_redone_by_hbm_build_of_k_phase": getattr(timed, "hbm_build_dnms", None)},
            "generate_s": round(t_gen, 1), "numa_node": numa,
        }
        if staged:
            out["link"] = {"bytes_per_step": int(staged["bytes"]), "read_records_staged": int(staged["records"]), "sites_staged": staged["sites"], "site_stage": staged["site_stage"],
                           "bytes_per_dnm": round(staged["bytes"] / n, 1),
                           "achieved_GBps": round(staged["bytes"] * args.steps / staged["elapsed"] / 1e9, 2), "peak_GBps": 64.0,
                           "chunks": staged["chunks"], "decode_s": round(staged["decode_s"], 1),
                           "result_mismatches_vs_resident": staged["mismatches_vs_resident"]}
            out["kernels_ms_per_step_staged"] = kern_ms(staged["prof"])
            if staged.get("streamed"):
                st_ = dict(staged["streamed"])
                st_["link_achieved_GBps"] = round(staged["bytes"] / (st_["ms_per_step"] * 1e-3) / 1e9, 2)
                out["streamed"] = st_
        print(json.dumps(out))
    wl.free()
    if dist is not None:
        dist.destroy_process_group()


def feed_e2e(args, eng, sc, ev, cl, cfg, P, mode, res_r, PinnedPool, dnms=None, filler=0.0, lite=False, cutoff_of_the_pile_ups=None):
    """Files -> results (what feeds the timed step).  The pile-ups of the first `--feed-dnms` DNMs are written as a
    real coordinate-sorted BAM + BAI (samtools layout, deflate level 6) and the WHOLE sites table as a BGZF VCF + TBI (synth/uzfiles.cpp:
    outside the timing).  Timed, per chunk of DNMs and with the real dependency inside the timer:
        the chunk's windows of the VCF through the tabix index (uz_vcf_decode_regions) -> site + genotype columns up, K1 + K2, het lists back
        -> the fetches they imply (staging.fetch_points) -> the BAM through its BAI straight into the link form in pinned memory
        (uz_bam_stage_*: inflate, walk, fetch reach + mates, pack) -> upload -> read stage -> results back.
    Host work of chunk k + 1 (VCF windows) and of chunk k (BAM stage) runs on worker threads beside the device work of chunk k - 1.
    `value_e2e` = DNMs / that wall time; the results are compared with the resident pass (same DNMs)."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from synth import bigsynth
    from unfazed_amd import abi, io_native
    from unfazed_amd.hostpath import concordant_cutoff
    from unfazed_amd.staging import fetch_points
    # dnms / filler / lite: the `feed_filler` leg -- fewer DNMs, the gaps between their pile-ups filled with read pairs at `filler`-fold coverage
    # (bigsynth.write_bam(filler=...)), and only the pass itself timed (no product call, no CPU decode)
    m_want = min(dnms if dnms is not None else args.feed_dnms, ev.n)
    c_hi = cl.of_dnm(m_want - 1) + 1
    m = int(cl.d0[c_hi - 1] + cl.nd[c_hi - 1])  # whole clusters
    base = args.feed_dir
    if base is None:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (64 << 30) else None
    d = tempfile.mkdtemp(prefix="uzfeed_", dir=base)
    try:
        t0 = time.perf_counter()
        bam = os.path.join(d, "kid.bam")
        st_b = bigsynth.write_bam(bam, cfg, sc, ev, cl, 0, c_hi, level=args.feed_level, tags=True, threads=0, filler=float(filler), filler_reach=65536)
        t1 = time.perf_counter()
        vcf = os.path.join(d, "sites.vcf.gz")
        st_v = bigsynth.write_vcf(vcf, sc, level=args.feed_level, threads=0)
        t2 = time.perf_counter()
        src = io_native.BamSource(bam, threads=0)
        names = io_native.tabix_contigs(vcf)
        tbx = {nm: i for i, nm in enumerate(names)}
        bam_tid = {nm: i for i, nm in enumerate(src.contigs)}
        cutoff = concordant_cutoff(src.tlen_head, P.readlen, 3)
        cutoff_file = cutoff
        if filler and cutoff_of_the_pile_ups is not None:
            # The insert cutoff is an estimate from the HEAD of the alignment file (read_collector.py:11-25): a file with filler in front of its first
            # pile-up has another head, its estimate differs from the pile-ups' in the third digit, and a handful of reads near the cutoff change sides.
            # The pass is held against the resident results, so it phases with THEIR cutoff; both are in the line.
            cutoff = float(cutoff_of_the_pile_ups)
        sd = int(P.search_dist) + 2
        cuts = list(range(0, m, max(1, args.feed_chunk))) + [m]
        if args.feed_walk == "device" and args.feed_inflate == "device" and len(cuts) > 3:
            # device walk: the device's share of a chunk (blocks up, inflate, walk) is the pipeline's longest stage, so the pass is its sum plus what
            # cannot overlap -- the first chunk's way to the device, the last chunk's joins and read stage: half-size chunks at both ends
            c = max(1, args.feed_chunk)
            cuts = [0, c // 2] + list(range(c // 2 + c, m - c // 2, c))
            if m - cuts[-1] > c // 2 + c // 4:
                cuts.append(m - c // 2)
            cuts.append(m)
        K = len(cuts) - 1
        NB = 4  # sets of page-locked buffers: chunk k uses set k % NB (up to three BAM stages in flight beside the read stage of a fourth chunk)
        pools = [PinnedPool() for _ in range(NB)]  # chunk k stages into pools[k % NB]: its block is rewound, not re-pinned
        from unfazed_amd.engine import PinnedPair
        on_device = args.feed_inflate == "device"
        dev_walk = args.feed_walk == "device" and on_device
        dev_joins = dev_walk and args.feed_joins == "device"
        ipairs = [PinnedPair() for _ in range(NB)] if on_device else None  # ... and its gathered / inflated BGZF blocks through ipairs[k % NB]
        slab_bytes = [0] * NB
        acc = dict(vcf_s=0.0, bam_s=0.0, site_records=0, walked=0, kept=0, file_bytes=0, blocks=0, spans=0.0, walk=0.0, mates=0.0, numbering=0.0, fill=0.0,
                   link_bytes=0, lookups=0, dev_blocks=0, dev_out_bytes=0, dev_inflate_s=0.0, gather_s=0.0, blocks_dev=0,
                   w_plan=0.0, w_walk=0.0, w_joins=0.0, w_kept=0.0, w_desc=0, w_host_tasks=0, w_tasks=0, w_aux=0, w_join_calls=0)

        def io_dev(packed):
            return packed.io_stats.get("blocks_from_the_device", 0)
        out = dict(status=np.empty(m, np.int32), counts=np.empty((m, 4), np.int32), origin=np.empty(m, np.int32), evidence=np.empty(m, np.int32))

        def stage_a(k):  # the chunk's windows of the sites file -> columns
            a, b = cuts[k], cuts[k + 1]
            t = time.perf_counter()
            cn = [sc.contig_names[c] for c in ev.contig[a:b]]
            tb = io_native.read_vcf_table_regions(vcf, [tbx[x] for x in cn], np.maximum(ev.start[a:b].astype(np.int64) - sd, 0), ev.end[a:b].astype(np.int64) + sd, threads=0)
            fam = tb.family_columns("kid", "dad", "mom")
            acc["vcf_s"] += time.perf_counter() - t
            acc["site_records"] += int(tb.pos.size)
            return tb, fam, cn

        def stage_b(k, got):  # device: site stage; host: the fetches it implies
            a, b = cuts[k], cuts[k + 1]
            tb, fam, cn = got
            sid = eng.upload_sites(tb)
            fid = eng.add_family(sid, *fam)
            rc = np.array([bam_tid[x] for x in cn], np.int32)
            dv = abi.dnms_view([tb.contig_index[x] for x in cn], rc, ev.start[a:b], ev.end[a:b], np.zeros(b - a, np.uint8), ev.refs[a:b], ev.alts[a:b], cutoff)
            co, ci, cf, ho, hi = eng.find(fid, dv, P, mode)
            alen = np.array([max(len(r), len(x)) for r, x in zip(ev.refs[a:b], ev.alts[a:b])], np.int64)
            f = fetch_points(rc, ev.start[a:b], np.zeros(b - a, np.uint8), tb.pos, ho, hi, P, allele_len=alen)
            return sid, fid, dv, f

        def stage_c(k, f):  # the BAM through its index, straight into the link form (pinned)
            t = time.perf_counter()
            pool = pools[k % NB]
            want = max(slab_bytes) * 5 // 4 if max(slab_bytes) else (cuts[k + 1] - cuts[k]) * 16384 + (1 << 20)
            if not pool.rewind(want):
                pool.free_all()
                pool.new_slab(want)
            if on_device:
                ipairs[k % NB].start()
            if dev_walk and dev_joins:  # the blocks go up, are inflated, walked AND joined in HBM: down come the walk tasks' flags and the table's totals
                pa = ipairs[k % NB].alloc
                kb = src.select_kept(f[0], f[1], f[2], int(P.min_gt_qual), join=eng, alloc=pa, release=eng.bam_walk_release)
                acc["bam_s"] += time.perf_counter() - t
                io, tm = kb.io_stats, kb.timing
                acc["walked"] += io["records_walked"]; acc["kept"] += io["records_kept"]; acc["lookups"] += io["index_mate_lookups"]
                acc["dev_blocks"] += kb.plan["n_blocks"]; acc["dev_out_bytes"] += kb.plan["out_bytes"]; acc["file_bytes"] += kb.plan["comp_bytes"]
                acc["w_plan"] += tm["plan"]; acc["w_walk"] += tm["walk"]; acc["w_joins"] += tm["joins"]
                acc["w_desc"] += int(kb.n_desc); acc["w_host_tasks"] += int(kb.host_tasks); acc["w_tasks"] += int(kb.plan["task"].shape[0]); acc["w_aux"] += int(kb.n_aux)
                acc["w_join_calls"] += int(kb.join_calls)
                plan_bytes = sum(int(v.nbytes) for key, v in kb.plan.items() if isinstance(v, np.ndarray) and key != "comp")
                acc["link_bytes"] += int(kb.plan["comp_bytes"]) + plan_bytes + int(kb.n_extra_desc) * 64 + int(kb.n_aux) + int(kb.plan["task"].shape[0]) * 12
                kb.plan = None
                return kb
            if dev_walk:  # the blocks go up, are inflated and walked in HBM; descriptors come back, the joins run here, the kept list goes up
                pa = ipairs[k % NB].alloc
                kb = src.select_kept(f[0], f[1], f[2], int(P.min_gt_qual), walk=lambda plan: eng.bam_walk(plan, alloc=pa), alloc=pa, release=eng.bam_walk_release)
                acc["bam_s"] += time.perf_counter() - t
                io, tm = kb.io_stats, kb.timing
                acc["walked"] += io["records_walked"]; acc["kept"] += io["records_kept"]; acc["lookups"] += io["index_mate_lookups"]
                acc["dev_blocks"] += kb.plan["n_blocks"]; acc["dev_out_bytes"] += kb.plan["out_bytes"]; acc["file_bytes"] += kb.plan["comp_bytes"]
                acc["w_plan"] += tm["plan"]; acc["w_walk"] += tm["walk"]; acc["w_joins"] += tm["joins"]; acc["w_kept"] += tm["kept"]
                acc["mates"] += tm["mates"]; acc["numbering"] += tm["numbering"]
                acc["w_desc"] += int(kb.desc.size); acc["w_host_tasks"] += int(kb.host_tasks); acc["w_tasks"] += int(kb.d_first.size - 1); acc["w_aux"] += int(kb.n_aux)
                acc["link_bytes"] += int(kb.plan["comp_bytes"]) + int(kb.desc.size) * 64 + int(kb.n) * 32 + int(kb.n_aux)
                kb.plan = kb.desc = None
                return kb
            packed = src.select(f[0], f[1], f[2], int(P.min_gt_qual), alloc=pool.alloc, extra=f[3],
                                inflate=eng.inflate_blocks if on_device else None, inflate_alloc=ipairs[k % NB].alloc if on_device else None)
            if packed.pre_inflate:
                acc["dev_blocks"] += packed.pre_inflate["blocks"]; acc["dev_out_bytes"] += packed.pre_inflate["out_bytes"]
                acc["dev_inflate_s"] += packed.pre_inflate["inflate_s"]; acc["gather_s"] += packed.pre_inflate["gather_s"]
                acc["blocks_dev"] += io_dev(packed)
            slab_bytes[k % NB] = max(slab_bytes[k % NB], pool.slab_used())
            acc["bam_s"] += time.perf_counter() - t
            io, tm = packed.io_stats, packed.timing
            acc["walked"] += io["records_walked"]; acc["kept"] += io["records_kept"]; acc["file_bytes"] += io["file_bytes_read"]
            acc["blocks"] += io["blocks_inflated"]; acc["lookups"] += io["index_mate_lookups"]
            for key in ("spans", "walk", "mates", "numbering", "fill"):
                acc[key] += tm[key]
            acc["link_bytes"] += sum(int(x.nbytes) for x in packed.arrays.values())
            return packed

        def stage_d(k, dev, packed):  # upload + read stage + results
            a, b = cuts[k], cuts[k + 1]
            sid, fid, dv, _ = dev
            rid = eng.reads_from_bam(packed) if dev_walk else eng.upload_reads_packed(packed)
            rr = eng.phase_raw(fid, rid, dv, P, mode)
            for key in out:
                out[key][a:b] = rr[key]
            eng.free_reads(rid)
            eng.free_sites(sid)

        def run_pass():
            for key in acc:
                acc[key] = type(acc[key])()
            # device walk: the BAM stages of THREE chunks in flight -- the device's share of chunk k (blocks up, inflate, walk, descriptors down) runs
            # beside the host's joins of chunks k - 1 and k - 2 (four walked batches may wait on the device: uz_bam_walk)
            lag = int(os.environ.get("UZ_FEED_LAG", "3")) if dev_walk else 1
            with ThreadPoolExecutor(lag + 1) as ex:
                t = time.perf_counter()
                c_t = time.process_time()
                fa = {0: ex.submit(stage_a, 0)}
                fcs, devs = {}, {}
                for k in range(K):
                    if k + 1 < K:
                        fa[k + 1] = ex.submit(stage_a, k + 1)
                    devs[k] = stage_b(k, fa.pop(k).result())
                    fcs[k] = ex.submit(stage_c, k, devs[k][3])
                    if k >= lag:
                        stage_d(k - lag, devs.pop(k - lag), fcs.pop(k - lag).result())
                for k in range(max(0, K - lag), K):
                    stage_d(k, devs.pop(k), fcs.pop(k).result())
                eng.sync()
                acc["host_cpu_s"] = time.process_time() - c_t  # (the library's worker threads included)
                return time.perf_counter() - t

        def throttled():  # the cgroup's CPU throttle: periods in which the quota ran out, and the time waited (cpu.stat)
            try:
                kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
                return int(kv.get("nr_throttled", 0)), int(kv.get("throttled_usec", 0))
            except Exception:
                return 0, 0

        per_pass = []

        def watched_pass():
            a0, t0_ = eng.walk_slot_stats()["allocations"], throttled()
            el_ = run_pass()
            t1_ = throttled()
            per_pass.append({"seconds": round(el_, 3), "device_allocations_by_the_walk_slots": eng.walk_slot_stats()["allocations"] - a0,
                             "cgroup_throttled_periods": t1_[0] - t0_[0], "cgroup_throttled_ms": round((t1_[1] - t0_[1]) / 1e3, 1)})
            return el_
        run_pass()  # warm-up: page cache, pinned blocks, code
        els = sorted(watched_pass() for _ in range(max(1, args.feed_reps)))
        el = els[len(els) // 2]  # the median pass (the stage statistics below are the last pass's)
        for pool in pools:
            pool.free_all()
        for ip in ipairs or []:
            ip.free_all()
        mism = sum(int((np.asarray(out[k]) != np.asarray(res_r[k][:m])).sum()) for k in out)
        if lite:
            return {
                "dnms": m, "chunks": K, "filler_coverage": float(filler), "insert_cutoff_used": round(float(cutoff), 3), "insert_cutoff_of_this_files_head": round(float(cutoff_file), 3),
                "value_e2e": round(m / el, 1), "seconds": round(el, 3),
                "seconds_of_every_pass": [round(x, 3) for x in els], "result_mismatches_vs_resident": mism,
                "bam": {"records": st_b["records"], "file_GB": round(st_b["file_bytes"] / 1e9, 3), "raw_GB": round(st_b["raw_bytes"] / 1e9, 3), "blocks": st_b["blocks"],
                        "write_s": round(t1 - t0, 1)},
                "blocks_inflated_per_dnm": round(acc["dev_blocks"] / m, 2), "inflated_MB_per_dnm": round(acc["dev_out_bytes"] / m / 1e6, 3),
                "records_walked_per_record_kept": round(acc["walked"] / max(1, acc["kept"]), 2), "records_kept_per_dnm": round(acc["kept"] / m, 1),
                "descriptors_left_in_hbm_per_dnm": round(acc["w_desc"] / m, 1), "tasks": acc["w_tasks"], "tasks_walked_by_the_host": acc["w_host_tasks"],
                "index_mate_lookups": acc["lookups"], "link_bytes_per_dnm": round(acc["link_bytes"] / m, 1),
                "host_cpu_seconds_per_pass": round(float(acc.get("host_cpu_s", 0.0)), 3),
                "seconds_busy": {"plan+gather": round(acc["w_plan"], 3), "upload+inflate+walk": round(acc["w_walk"], 3), "joins": round(acc["w_joins"], 3),
                                 "sites_decode": round(acc["vcf_s"], 3)},
            }
        product = product_e2e(bam, vcf, sc, ev, m, res_r) if not os.environ.get("UZ_BENCH_NO_PRODUCT") else None
        # the CPU path's decode of the same files: the whole BAM (it holds only these pile-ups) + the same windows of the VCF
        t = time.perf_counter()
        full = io_native.read_bam_table(bam, threads=0)
        cpu_records = int(full.start.size)
        del full
        cn = [sc.contig_names[c] for c in ev.contig[:m]]
        io_native.read_vcf_table_regions(vcf, [tbx[x] for x in cn], np.maximum(ev.start[:m].astype(np.int64) - sd, 0), ev.end[:m].astype(np.int64) + sd, threads=0)
        cpu_decode_s = time.perf_counter() - t
        raw_per_rec = st_b["raw_bytes"] / max(1, st_b["records"])
        return {
            "dnms": m, "chunks": K, "value_e2e": round(m / el, 1), "seconds": round(el, 3), "seconds_of_every_pass": [round(x, 3) for x in els],
            "every_pass": per_pass,
            "result_mismatches_vs_resident": mism,
            "walk": ("device (k_bam_walk: one wavefront per walk task; mate() closure, name numbering, file order and the table's offsets on the device too: csrc/k_bamjoin.hip)" if dev_joins
                     else "device (k_bam_walk: one wavefront per walk task; the host runs the batch-wide joins on 64-byte descriptors)") if dev_walk else "host (uz_bam_stage_*)",
            "joins": "device" if dev_joins else "host",
            "inflate": ("device (k_bgzf_inflate: one wavefront per BGZF block; blocks the walk did not announce: " + io_native.inflate_backend() + ")") if on_device
                       else io_native.inflate_backend(),
            "device_inflate": None if not on_device else {"blocks": acc["dev_blocks"], "out_GB": round(acc["dev_out_bytes"] / 1e9, 3),
                                                          "note": "inflated in HBM and walked there: its time is inside device_walk.seconds_busy"} if dev_walk else {
                "blocks": acc["dev_blocks"], "blocks_used_by_the_walk": acc["blocks_dev"], "out_GB": round(acc["dev_out_bytes"] / 1e9, 3),
                "seconds_busy": round(acc["dev_inflate_s"], 3), "gather_seconds_busy": round(acc["gather_s"], 3),
                "GBps_incl_both_copies": round(acc["dev_out_bytes"] / max(acc["dev_inflate_s"], 1e-9) / 1e9, 1),
                "note": "per chunk: the blocks the walk will read gathered into pinned memory (host), host -> device, kernel, device -> host, all "
                        "inside seconds_busy; the walk then copies records out of the inflated blocks and holds each block against its CRC-32"},
            "device_walk": None if not dev_walk else {
                "kernels": "k_bgzf_inflate + k_bgzf_crc32, k_bam_walk (one wavefront per walk task), k_tab_insert + k_desc_filter, k_bam_extract (csrc/k_inflate.hip, k_bamwalk.hip)", "tasks": acc["w_tasks"],
                "tasks_walked_by_the_host": acc["w_host_tasks"], "descriptors": acc["w_desc"], "aux_bytes": acc["w_aux"],
                "seconds_busy": ({"plan+gather": round(acc["w_plan"], 3), "upload+inflate+walk": round(acc["w_walk"], 3),
                                  "joins_on_the_device (flags down, the host's own walks, uz_bam_join until it needs nothing)": round(acc["w_joins"], 3),
                                  "joins_on_the_host": 0.0, "kept_list": 0.0} if dev_joins else
                                 {"plan+gather": round(acc["w_plan"], 3), "upload+inflate+walk+descriptors_down": round(acc["w_walk"], 3),
                                  "joins_on_the_host": round(acc["w_joins"], 3), "kept_list": round(acc["w_kept"], 3)}),
                "join_calls": acc["w_join_calls"] if dev_joins else None,
                "link_bytes_per_dnm": round(acc["link_bytes"] / m, 1),
                "note": ("the inflated bytes (out_GB) and the descriptors never cross the link: up go the compressed blocks, the walk plan and what the host walked itself "
                         "(tasks handed back, mates looked up through the index: descriptors + bytes), down come the walk tasks' flags and the table's totals") if dev_joins else
                        ("the inflated bytes (out_GB) never cross the link: up go the compressed blocks and the kept list (32 B per record), down come the "
                         "descriptors (64 B per record inside a reach interval)")},
            "host_threads": io_native.default_threads(), "host_cpu_quota": io_native.cpu_quota() or None,
            "host_processors": os.cpu_count(),
            "bam": {"records": st_b["records"], "file_GB": round(st_b["file_bytes"] / 1e9, 3), "raw_GB": round(st_b["raw_bytes"] / 1e9, 3), "blocks": st_b["blocks"],
                    "write_s": round(t1 - t0, 1), "deflate_level": args.feed_level},
            "vcf": {"records": st_v["records"], "file_GB": round(st_v["file_bytes"] / 1e9, 3), "write_s": round(t2 - t1, 1)},
            # (rates only for a stage that ran on the host: with the walk on the device its time is inside device_walk.seconds_busy)
            "decode_regions": dict({"seconds_busy": round(acc["walk"], 3), "records_walked": acc["walked"], "blocks_inflated": acc["blocks"],
                                    "file_bytes_read": acc["file_bytes"], "index_mate_lookups": acc["lookups"]},
                                   **({"records_walked_per_s": round(acc["walked"] / acc["walk"], 0),
                                       "inflated_GBps": round(acc["walked"] * raw_per_rec / acc["walk"] / 1e9, 2)} if acc["walk"] > 1e-6 else {})),
            "host_cpu_seconds_per_pass": round(float(acc.get("host_cpu_s", 0.0)), 3),
            "select": {"seconds_busy": round(acc["spans"] + acc["mates"], 3), "spans_s": round(acc["spans"], 3), "mates_s": round(acc["mates"], 3), "records_kept": acc["kept"],
                       "records_per_s": round(acc["kept"] / max(acc["spans"] + acc["mates"], 1e-9), 0)},
            "pack": {"seconds_busy": round(acc["numbering"] + acc["fill"], 3), "numbering_s": round(acc["numbering"], 3), "fill_s": round(acc["fill"], 3), "records_per_s": round(acc["kept"] / max(acc["numbering"] + acc["fill"], 1e-9), 0),
                     "link_bytes_per_dnm": round(acc["link_bytes"] / m, 1)},
            "sites_decode": {"seconds_busy": round(acc["vcf_s"], 3), "records": acc["site_records"], "records_per_s": round(acc["site_records"] / max(acc["vcf_s"], 1e-9), 0)},
            "bam_stage_seconds_busy": round(acc["bam_s"], 3),
            "product": product,
            "cpu_decode_s": round(cpu_decode_s, 3), "cpu_decode_records": cpu_records,
            "note": "files -> BED-ready results with the real dependency find(k) -> fetches(k) -> BAM stage(k) -> upload(k) inside the timer; the BAM holds the "
                    "+-6 kb pile-ups of the DNMs only (no filler between windows), so a window's BGZF blocks carry no lead-in from a 16 kb index bin",
        }
    finally:
        shutil.rmtree(d, ignore_errors=True)


def product_e2e(bam, vcf, sc, ev, m, res_r):
    """The same files through the PRODUCT's drop-in call -- phase_snvs(dnms, kids, pedigrees, sites, ...) as the reference's driver makes it
    (unfazed.py:601-646) -> session -> hostpath (chunks of DNMs, chunk k + 1 decoded while chunk k is phased) -> records dict -- timed on the
    three calls after a first one (the median is reported), and its records held against the resident pass (read and site counts of every DNM)."""
    from unfazed_amd import abi, session
    from unfazed_amd.snv_phaser import phase_snvs
    ped = {"kid": {"kid": "kid", "dad": "dad", "mom": "mom", "sex": "2"}}
    dnms = [dict(chrom=sc.contig_names[int(c)], start=int(s), end=int(e), kid="kid", vartype="POINT", bam=bam, cram_ref=None)
            for c, s, e in zip(ev.contig[:m], ev.start[:m], ev.end[:m])]
    argv = (["kid"], ped, vcf, 2, "38", False, 10 ** 9, True, [0.0, 0.2], [0.8, 1.0], [0.2, 0.8], 20, 10, 5000, 1000000, 3, 1, 151, 5)
    el, recs, els = 0.0, {}, []
    for rep in range(4):  # (one warm call, then three timed ones: the median is reported, as for the feed pass)
        session._HOSTS.clear()
        for k in [k for k in session._SITES if "@" in k]:
            del session._SITES[k]
        batch = [dict(x) for x in dnms]
        prof = None
        if rep == 1 and os.environ.get("UZ_BENCH_PROFILE_PRODUCT"):  # development aid: where the drop-in call's host time goes
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t = time.perf_counter()
        recs = phase_snvs(batch, *argv)
        el = time.perf_counter() - t
        if rep >= 1:
            els.append(el)
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(40)
    # (the sites file spells every DNM's own record as an SNV, so the driver -- which takes REF / ALT from that file, snv_phaser.py:73-84 -- phases
    # the batch's INDEL DNMs as SNVs: only the SNV DNMs are the same question in both passes)
    # (and a DNM with a second site record at its position or the base before gets "Too many genotypes" from the driver's REF / ALT look-up over
    # 1-based [pos, pos + 1], snv_phaser.py:73-84, :117-130 -- the read stage is never asked: not a question both passes answer either)
    bad, compared, refalt_skips = 0, 0, 0
    st, cnt = np.asarray(res_r["status"][:m]), np.asarray(res_r["counts"][:m])
    co_s = np.asarray(sc.contig_off, np.int64)
    for d in range(m):
        if ev.kind[d] != 0:
            continue
        c, p0 = int(ev.contig[d]), int(ev.start[d])
        seg = sc.pos[co_s[c]: co_s[c + 1]]
        if int(np.searchsorted(seg, p0, "right") - np.searchsorted(seg, p0 - 1, "left")) != 1:
            refalt_skips += 1
            continue
        compared += 1
        key = "%s_%d_%d_kid_POINT" % (dnms[d]["chrom"], dnms[d]["start"], dnms[d]["end"])
        r = recs.get(key)
        if st[d] != abi.ST_OK:
            bad += r is not None
        else:
            bad += r is None or [len(r["dad_reads"]), len(r["mom_reads"]), len(r["dad_sites"]), len(r["mom_sites"])] != cnt[d].tolist()
    el = sorted(els)[len(els) // 2]
    return {"value_e2e": round(m / el, 1), "seconds": round(el, 3), "seconds_of_every_call": [round(x, 3) for x in els], "records": len(recs), "snv_dnms_compared": compared, "dnms_the_driver_skips_at_its_refalt_lookup": refalt_skips, "record_mismatches_vs_resident": int(bad),
            "route": "phase_snvs -> session -> hostpath._chunked_batch (%d DNMs per chunk) -> HipEngine" % __import__("unfazed_amd.hostpath", fromlist=["x"]).PhasingHost.CHUNK_DNMS}


def cpu_baseline(args, wl, sc, ev, dn, per_ev, cl, cfg, P, cutoff, gpu_res, ev_vt, ev_refs, ev_alts, cnv, parity_only=False):
    """The CPU oracle (C port of the reference's algorithm) timed on this host, on the first `cpu_dnms` DNMs of the GPU's batch
    (whole clusters).  The sample is cut into contiguous cluster ranges, each with its OWN records table (regenerated on the
    host by the gcc build of the generator, outside the timing -- as separate workers each reading their own regions of the
    alignment file would hold it); T threads take the ranges round-robin, one oracle call (find + phase) per range.  The
    oracle's per-call set-up is proportional to the table it is handed, so one shared table does not scale past a few
    threads; per-range tables do.  Results are compared with the GPU's."""
    if cnv:  # the two breakpoints of an event lie in different clusters
        return cpu_baseline_cnv_ranges(args, sc, ev, dn, cl, cfg, P, cutoff, gpu_res, parity_only=parity_only)
    from oracle import oracle as orc
    from synth import bigsynth
    from unfazed_amd import abi
    ncpu = os.cpu_count() or 1
    c_hi = cl.of_dnm(per_ev * min(args.cpu_dnms, ev.n) - 1) + 1
    m = int(cl.d0[c_hi - 1] + cl.nd[c_hi - 1]) // per_ev  # whole clusters
    nc = len(sc.contig_off) - 1
    sv = abi.SitesView()
    keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=sc.pos, sflags=sc.sflags,
                ref_base=sc.ref_base, alt_base=sc.alt_base)
    sv.n_sites, sv.n_contigs = sc.n, nc
    for k, a in keep.items():
        setattr(sv, k, a.ctypes.data)
    sh = abi.Held(sv, keep)
    # the GPU folded the complex flag into bit 6 of its own copy of gt; the host copy is untouched
    fh = abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)
    orc.lib()
    # ranges: cluster boundaries nearest to an even split of the sample's entries of `dn` (DNMs / breakpoints)
    n_ranges = max(1, min(4 * max(ncpu, 64), c_hi))
    ends = cl.d0[:c_hi] + cl.nd[:c_hi]
    cuts = sorted({int(np.searchsorted(ends, per_ev * m * (k + 1) / n_ranges, side="left")) + 1 for k in range(n_ranges)} | {c_hi})
    cuts = [c for c in cuts if c <= c_hi]
    ranges, c0 = [], 0
    for c1 in cuts:
        if c1 > c0:
            ranges.append((c0, c1))
            c0 = c1
    jobs = []
    for (a, b) in ranges:
        e0 = int(cl.d0[a]) // per_ev
        e1 = min(m, int(cl.d0[b - 1] + cl.nd[b - 1]) // per_ev)
        if e1 <= e0:
            continue
        rh, _ = bigsynth.reads_cpu(cfg, sc, dn, cl, a, b, threads=min(ncpu, 64))
        dv = abi.dnms_view(ev.contig[e0:e1], ev.contig[e0:e1], ev.start[e0:e1], ev.end[e0:e1], ev_vt[e0:e1], ev_refs[e0:e1], ev_alts[e0:e1], cutoff)
        jobs.append((e0, e1, rh, dv))

    def run(threads):
        out = dict(status=np.full(m, abi.ST_SKIPPED, np.int32), counts=np.zeros((m, 4), np.int32), origin=np.zeros(m, np.int32),
                   evidence=np.zeros(m, np.int32))
        if cnv:
            out["etype"] = np.zeros(m, np.int32)
            out["cnv_counts"] = np.zeros((m, 2), np.int32)
        t0 = time.perf_counter()

        def work(t):
            for j in range(t, len(jobs), threads):
                e0, e1, rh, dv = jobs[j]
                found = orc.find(P, sh, fh, dv, abi.FIND_SECOND_WINDOW)
                p = orc.phase(P, sh, rh, dv, found, keep_lists=False)
                out["status"][e0:e1], out["counts"][e0:e1] = p["status"], p["counts"]
                if cnv:
                    k = orc.phase_cnv(P, sh, fh, dv, rb_counts=p["counts"])
                    for name in ("origin", "evidence", "etype", "cnv_counts"):
                        out[name][e0:e1] = k[name]
                else:
                    out["origin"][e0:e1], out["evidence"][e0:e1] = p["origin"], p["evidence"]

        th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0, out

    if parity_only:  # (tests: one pass on every core the box grants, no timing ladder)
        dtp, rp = run(_cpus())
        dt1, r1, dt2 = dtp, rp, dtp
        best, sweep = (dtp, rp, _cpus()), {str(_cpus()): round(m / dtp, 1)}
    else:
        dt1, r1 = run(1)
        dt2, _ = run(2)
        # thread counts up to twice the CPUs the cgroup grants (threads beyond the quota only take turns: round 3's "best at 256 threads" on a
        # 16-CPU quota was noise)
        ladder = sorted({t for t in (8, _cpus(), 2 * _cpus()) if 2 < t <= ncpu})
        best = None
        sweep = {"1": round(m / dt1, 1), "2": round(m / dt2, 1)}
        for t in ladder:
            dtt, rt = run(t)
            sweep[str(t)] = round(m / dtt, 1)
            if best is None or dtt < best[0]:
                best = (dtt, rt, t)
        if best is None:
            best = (dt2, r1, 2) if dt2 < dt1 else (dt1, r1, 1)
    dtc, rc, threads_best = best
    cores = min(threads_best, _cpus())  # the CPUs those threads really ran on
    mism = 0
    for k in r1:
        mism += int((np.asarray(gpu_res[k][:m]) != r1[k]).sum())
        mism += int((rc[k] != r1[k]).sum())
    return {"value": round(m / dtc, 1), "unit": "DNMs/s", "cores": cores, "kind": "port",
            "threads": threads_best,
            "sample": "first %d DNMs of the GPU batch (whole read clusters) in %d cluster ranges, each with its own records table regenerated on the host "
                      "before the timing; oracle find+phase per range, threads over ranges (best of a ladder of thread counts: %d threads on %d CPUs)" % (m, len(jobs), threads_best, cores),
            "value_1thread": round(m / dt1, 1), "value_2threads": round(m / dt2, 1),
            "seconds": {"1": round(dt1, 2), "2": round(dt2, 2), str(threads_best): round(dtc, 2)},
            "dnms_per_s_by_threads": sweep, "host_hw_threads": ncpu, "host_cpu_quota": _cpu_quota(),
            "parity_mismatches_vs_gpu": mism}


def _cpus():
    """CPUs this process can really run on at once: the cgroup's quota when there is one (a pod that lists 256 processors may be held to 16),
    else the processors listed"""
    q = _cpu_quota()
    n = os.cpu_count() or 1
    return max(1, min(n, int(q))) if q else n


def _cpu_quota():
    """CPUs the container's cgroup grants (cpu.max), None = unlimited: a pod that lists 256 processors may be held to 16"""
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if a == "max" else round(int(a) / int(b), 2)
    except Exception:
        return None


def cpu_baseline_cnv_ranges(args, sc, ev, dn, cl, cfg, P, cutoff, gpu_res, parity_only=False, n_ranges=None):
    """Config 5 on the CPU oracle with per-range tables, as the SNV baseline has them: the breakpoint list (sorted by position) is cut
    into contiguous cluster ranges, each with its own records table regenerated on the host before the timing; the sample is the
    events whose BOTH breakpoints fall into one range (an event that straddles a cut is left out: a few per cent); T threads take the
    ranges round-robin, one oracle pass (find + read stage + allele balance) per range.  Results are compared with the GPU's."""
    from oracle import oracle as orc
    from synth import bigsynth
    from unfazed_amd import abi
    ncpu = os.cpu_count() or 1
    want = min(args.cpu_dnms, ev.n)
    n_ranges = int(n_ranges) if n_ranges else max(1, min(4 * ncpu, cl.n // 8))
    ends = cl.d0 + cl.nd
    cuts = sorted({int(np.searchsorted(ends, dn.n * (k + 1) / n_ranges, side="left")) + 1 for k in range(n_ranges)} | {cl.n})
    cuts = [c for c in cuts if c <= cl.n]
    c_of_start = np.searchsorted(cl.d0, dn.bp_start, side="right") - 1
    c_of_end = np.searchsorted(cl.d0, dn.bp_end, side="right") - 1
    nc = len(sc.contig_off) - 1
    sv = abi.SitesView()
    keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=sc.pos, sflags=sc.sflags, ref_base=sc.ref_base, alt_base=sc.alt_base)
    sv.n_sites, sv.n_contigs = sc.n, nc
    for k, a in keep.items():
        setattr(sv, k, a.ctypes.data)
    sh = abi.Held(sv, keep)
    fh = abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)
    orc.lib()
    jobs, c0, taken = [], 0, 0
    for c1 in cuts:
        if c1 <= c0:
            continue
        inside = np.nonzero((c_of_start >= c0) & (c_of_start < c1) & (c_of_end >= c0) & (c_of_end < c1))[0]
        if inside.size and taken < want:
            inside = inside[: want - taken]
            rh, _ = bigsynth.reads_cpu(cfg, sc, dn, cl, c0, c1, threads=min(ncpu, 32))
            dv = abi.dnms_view(ev.contig[inside], ev.contig[inside], ev.start[inside], ev.end[inside], ev.vartype[inside], [b""] * inside.size, [b""] * inside.size, cutoff)
            jobs.append((inside, rh, dv))
            taken += int(inside.size)
        c0 = c1
    idx_all = np.concatenate([j[0] for j in jobs]) if jobs else np.zeros(0, np.int64)
    m = int(idx_all.size)

    def run(threads):
        out = {k: {} for k in ("status", "counts", "origin", "evidence", "etype", "cnv_counts")}
        t0 = time.perf_counter()

        def work(t):
            for j in range(t, len(jobs), threads):
                inside, rh, dv = jobs[j]
                found = orc.find(P, sh, fh, dv, abi.FIND_SECOND_WINDOW)
                p = orc.phase(P, sh, rh, dv, found, keep_lists=False)
                k = orc.phase_cnv(P, sh, fh, dv, rb_counts=p["counts"])
                out["status"][j], out["counts"][j] = p["status"], p["counts"]
                for name in ("origin", "evidence", "etype", "cnv_counts"):
                    out[name][j] = k[name]

        th = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        return dt, {k: np.concatenate([v[j] for j in range(len(jobs))]) for k, v in out.items()}

    if parity_only:  # (tests: one pass on every core the box grants, no timing ladder)
        dtp, rp = run(_cpus())
        dt1, r1, dt2 = dtp, rp, dtp
        best, sweep = (dtp, rp, _cpus()), {str(_cpus()): round(m / dtp, 1)}
    else:
        dt1, r1 = run(1)
        dt2, _ = run(2)
        # thread counts up to twice the CPUs the cgroup grants (threads beyond the quota only take turns: round 3's "best at 256 threads" on a
        # 16-CPU quota was noise)
        ladder = sorted({t for t in (8, _cpus(), 2 * _cpus()) if 2 < t <= ncpu})
        best = None
        sweep = {"1": round(m / dt1, 1), "2": round(m / dt2, 1)}
        for t in ladder:
            dtt, rt = run(t)
            sweep[str(t)] = round(m / dtt, 1)
            if best is None or dtt < best[0]:
                best = (dtt, rt, t)
        if best is None:
            best = (dt2, r1, 2) if dt2 < dt1 else (dt1, r1, 1)
    dtc, rc, threads_best = best
    cores = min(threads_best, _cpus())  # the CPUs those threads really ran on
    mism = 0
    for k in r1:
        mism += int((np.asarray(gpu_res[k])[idx_all] != r1[k]).sum())
        mism += int((rc[k] != r1[k]).sum())
    return {"value": round(m / dtc, 1), "unit": "events/s", "cores": cores, "kind": "port",
            "threads": threads_best,
            "sample": "%d events of the GPU batch whose two breakpoints fall into one of %d cluster ranges (each range with its own records table, regenerated on the "
                      "host before the timing); oracle find + read stage + allele balance per range, threads over ranges (best of a ladder: %d threads on %d CPUs)" % (m, len(jobs), threads_best, cores),
            "value_1thread": round(m / dt1, 1), "value_2threads": round(m / dt2, 1),
            "seconds": {"1": round(dt1, 2), "2": round(dt2, 2), str(threads_best): round(dtc, 2)},
            "dnms_per_s_by_threads": sweep, "host_hw_threads": ncpu, "host_cpu_quota": _cpu_quota(),
            "parity_mismatches_vs_gpu": mism}


if __name__ == "__main__":
    main()
