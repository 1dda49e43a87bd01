"""Host-side scaling probe of the one-pass BAM stage (uz_bam_stage_*) and the tabix region decode: writes a config-3-density
BAM + VCF for N DNMs and times the stage at several thread counts, phase by phase.  Runs anywhere (no GPU call).
    python scripts/feed_probe.py [N=10000] [threads,threads,...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from synth import bigsynth  # noqa: E402
from synth.sites_np import make_clusters, make_sites, place_dnms_full  # noqa: E402
from unfazed_amd import io_native  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ladder = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 32, 64, 128, 256]
L = int(31000 * N)  # 100 k DNMs on 3.1 Gb
sc = make_sites(int(L / 155), seed=202, contig_lens=[L])
dn = place_dnms_full(sc, N, seed=201)
cl = make_clusters(dn)
cfg = bigsynth.make_cfg(seed=203)
cfg.n_clusters = cl.n
d = "/dev/shm/uzprobe"
os.makedirs(d, exist_ok=True)
bam, vcf = d + "/k.bam", d + "/s.vcf.gz"
t = time.time()
st = bigsynth.write_bam(bam, cfg, sc, dn, cl, contig_len=[L], level=6)
print("bam", st, "%.1fs" % (time.time() - t), flush=True)
t = time.time()
sv = bigsynth.write_vcf(vcf, sc, contig_len=[L], level=6)
print("vcf", sv, "%.1fs" % (time.time() - t), flush=True)
kid_het = (sc.gt & 3) == 1
hp = sc.pos[kid_het]
a = np.searchsorted(hp, dn.start - 5000)
b = np.searchsorted(hp, dn.start + 5000, "right")
cnt = b - a
idx = np.repeat(a, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
p = np.concatenate([dn.start, hp[idx]])
tid = np.zeros(p.size, np.int32)
lo, hi = (p - 1).astype(np.int32), (p + 1).astype(np.int32)
ex = np.zeros(p.size, np.uint16)
print("fetches per DNM %.1f, clusters %d, inflate %s, cpus %d" % (p.size / N, cl.n, io_native.inflate_backend(), len(os.sched_getaffinity(0))), flush=True)
for th in ladder:
    src = io_native.BamSource(bam, threads=th)
    best = None
    for rep in range(3):
        t = time.time()
        got = src.select(tid, lo, hi, 20, extra=ex)
        dt = time.time() - t
        if best is None or dt < best[0]:
            best = (dt, got.timing, got.io_stats)
        del got
    dt, tm, io = best
    print("threads %3d: %.3fs = %.0f DNMs/s | %s | walked %d kept %d blocks %d" % (th, dt, N / dt, {k: round(v, 3) for k, v in tm.items()}, io["records_walked"], io["records_kept"],
                                                                                io["blocks_inflated"]), flush=True)
    t = time.time()
    tb = io_native.read_vcf_table_regions(vcf, np.zeros(N, np.int32), np.maximum(dn.start - 5002, 0), dn.start + 5003, threads=th)
    print("            vcf regions %.3fs, %d records" % (time.time() - t, tb.pos.size), flush=True)
    t = time.time()
    full = io_native.read_bam_table(bam, threads=th)
    print("            whole-file decode %.3fs %s" % (time.time() - t, {k: round(v, 2) for k, v in full.decode_seconds.items()}), flush=True)
    del full
import shutil
shutil.rmtree(d, ignore_errors=True)
