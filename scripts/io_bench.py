"""Decode throughput of the native BAM / VCF decoders next to the Python decoders (host only).
usage: python scripts/io_bench.py [n_dnms] [threads]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synth.small import SmallConfig, make_small  # noqa: E402
from tests.filesio import dump_dataset  # noqa: E402
from unfazed_amd import io_native  # noqa: E402
from unfazed_amd.io_bam import read_bam  # noqa: E402
from unfazed_amd.io_vcf import read_vcf  # noqa: E402
from unfazed_amd.model import ReadsTable, SitesTable  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_dnms = int(args[0]) if len(args) > 0 else 150
threads = int(args[1]) if len(args) > 1 else 0
with_python = "--no-python" not in sys.argv
with tempfile.TemporaryDirectory() as d:
    ds = make_small(SmallConfig(seed=1, n_dnms=n_dnms))
    paths = dump_dataset(ds, d)
    bam = list(paths["bams"].values())[0]
    sz = os.path.getsize(bam)
    t0 = time.time(); t = io_native.read_bam_table(bam, threads=threads); t1 = time.time()
    print("BAM %d records, %.1f MB compressed: native %.3f s (%s) = %.2f M records/s" % (
        t.n_segs, sz / 1e6, t1 - t0, ", ".join("%s %.3f" % kv for kv in t.decode_seconds.items()), t.n_segs / (t1 - t0) / 1e6))
    for th in (1, 8, 32, 0):
        t0 = time.time(); io_native.read_bam_table(bam, threads=th); t1 = time.time()
        print("    threads=%d: %.3f s = %.2f M records/s, %.0f MB/s of BAM" % (th, t1 - t0, t.n_segs / (t1 - t0) / 1e6, sz / 1e6 / (t1 - t0)))
    if with_python:
        t0 = time.time(); c, segs = read_bam(bam); ReadsTable.from_segments(segs, c); t1 = time.time()
        print("    python decoder %.2f s = %.3f M records/s" % (t1 - t0, t.n_segs / (t1 - t0) / 1e6))
    t0 = time.time(); s = io_native.read_vcf_table(paths["sites"], threads=threads); t1 = time.time()
    print("VCF %d sites x %d samples: native %.4f s = %.2f M sites/s" % (s.n_sites, len(s.samples), t1 - t0, s.n_sites / (t1 - t0) / 1e6))
    if with_python:
        t0 = time.time(); smp, recs, _ = read_vcf(paths["sites"]); SitesTable.from_records(recs, smp); t1 = time.time()
        print("    python decoder %.3f s = %.3f M sites/s" % (t1 - t0, s.n_sites / (t1 - t0) / 1e6))
