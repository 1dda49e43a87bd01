"""Development aid (GPU box): the batch-wide joins on the device (csrc/k_bamjoin.hip) on fresh synth.small seeds -- pile-ups written as BAM + BAI by
the Python writer with the odd records of the golden sets mixed per seed (duplicates, secondary / supplementary copies, mates unmapped or on other
contigs, read lengths of 100 ... 2 500), random fetch sets, a reach slack of 0 ... 1 000 (mates through the index), walk tasks handed back to the host at
random -- against the one-pass stage (uz_bam_stage_plan: the host's joins) on the same fetches: same records, order, name ids, mates, bases for the
same records, per-reference counts.  usage: join_fuzz_gpu.py FIRST_SEED N"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from filesio import dump_dataset, write_bai  # noqa: E402
from synth.small import SmallConfig, make_small  # noqa: E402
from unfazed_amd import io_native  # noqa: E402
from unfazed_amd.engine import HipEngine  # noqa: E402


def variants(rng):
    cfg = dict(n_dnms=int(rng.randint(5, 12)), cluster_prob=float(rng.choice([0.3, 0.6, 1.0])))
    if rng.rand() < 0.5:
        cfg.update(odd_read_prob=float(rng.choice([0.1, 0.25])), softclip_prob=0.05, indel_prob=0.03)
    if rng.rand() < 0.3:
        cfg.update(lowq_prob=0.08)
    r = rng.rand()
    if r < 0.15:
        cfg.update(readlen=100)
    elif r < 0.3:
        cfg.update(readlen=301)
    elif r < 0.4:
        cfg.update(readlen=2500, coverage_per_hap=4.0, ins_mean=7500, ins_sd=100)
    env = {}
    if rng.rand() < 0.5:
        env["UZ_STAGE_SLACK"] = str(int(rng.choice([0, 30, 60, 300])))
    if rng.rand() < 0.4:
        env["UZ_TEST_FLAG_EVERY"] = str(int(rng.choice([2, 3, 5])))
    return cfg, env, int(rng.randint(0, 10)), bool(rng.rand() < 0.2)


first, n = int(sys.argv[1]), int(sys.argv[2])
eng = HipEngine(0)
bad = recs = lookups = handed = 0
for seed in range(first, first + n):
    rng = np.random.RandomState(seed)
    cfg, env, n_het, all_bases = variants(rng)
    ds = make_small(SmallConfig(seed=seed, **cfg))
    d = tempfile.mkdtemp(prefix="uzjf_")
    try:
        bam = list(dump_dataset(ds, d)["bams"].values())[0]
        write_bai(bam)
        full = io_native.read_bam_table(bam, threads=2)
        c, lo, hi = [], [], []
        for dn in ds.dnms:
            tid = full.contig_index[dn["chrom"]]
            c.append(tid); lo.append(dn["start"] - 1); hi.append(dn["start"] + 1)
            for p in np.sort(rng.randint(dn["start"] - 5000, dn["start"] + 5000, n_het)):
                c.append(tid); lo.append(int(p)); hi.append(int(p) + 1)
        fc, flo, fhi = np.array(c, np.int32), np.array(lo, np.int32), np.array(hi, np.int32)
        old = {k: os.environ.get(k) for k in ("UZ_STAGE_SLACK", "UZ_TEST_FLAG_EVERY")}
        for k in old:
            os.environ.pop(k, None)
        os.environ.update({k: v for k, v in env.items() if k == "UZ_STAGE_SLACK"})
        src = io_native.BamSource(bam, threads=3)
        ref = src.select(fc, flo, fhi, 20, all_bases=all_bases)
        k = int(ref.view.n_segs)
        voff, qn, mt, bs = io_native.stage_kept_debug(src.lib, ref._stage.ptr, k)
        os.environ.update(env)
        dev = src.select_kept(fc, flo, fhi, 20, all_bases=all_bases, join=eng, release=eng.bam_walk_release)
        got = eng.join_fetch(dev.token, dev.n, len(src.contigs))
        ok = (dev.n == k and np.array_equal(got["voff"], voff) and np.array_equal(got["qname"], qn) and np.array_equal(got["mate"], mt)
              and np.array_equal(got["bases"], bs) and dev.n_qnames == int(ref.view.n_qnames)
              and np.array_equal(got["contig_off"], ref.arrays["contig_off"][: got["contig_off"].size]))
        rid = eng.reads_from_bam(dev)  # (and the table builds from the list)
        eng.free_reads(rid)
        for key, v in old.items():
            os.environ.pop(key, None)
            if v is not None:
                os.environ[key] = v
        bad += not ok
        recs += k
        lookups += dev.io_stats["index_mate_lookups"]
        handed += dev.host_tasks
        if not ok or seed % 20 == 0:
            print("join seed %d: %d kept records, %d look-ups, %d tasks handed back, %s %s %s" % (seed, k, dev.io_stats["index_mate_lookups"], dev.host_tasks,
                                                                                                "ok" if ok else "MISMATCH", cfg, env), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
print("join fuzz: %d seeds, %d kept records, %d mates through the index, %d tasks handed back, %d mismatching seeds" % (n, recs, lookups, handed, bad))
sys.exit(1 if bad else 0)
