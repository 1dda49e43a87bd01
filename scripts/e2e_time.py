"""End-to-end timing from files on disk to the BED text, stage by stage (GPU box), for the two ways the session reads a BAM:
  whole   the whole file is inflated and decoded once (no index next to it), then phase_snvs on the resident table
  index   a BAI sits next to the BAM: per batch only the blocks its fetches (+ mates) live in are inflated
          (uz_bam_decode_regions), packed and uploaded
usage: python scripts/e2e_time.py [n_dnms] [spacing]      spacing: distance between DNMs (default 30000: reads cover ~40 % of
the file's span; the BAM only holds reads within +-6 kb of the DNMs, so even the whole-file decode sees no filler)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from filesio import dump_dataset, write_bai  # noqa: E402
from synth.small import SmallConfig, make_small  # noqa: E402
from unfazed_amd import io_native, session  # noqa: E402
from unfazed_amd.snv_phaser import phase_snvs  # noqa: E402
from unfazed_amd.unfazed import write_bed_output  # noqa: E402

n_dnms = int(sys.argv[1]) if len(sys.argv) > 1 else 400
with tempfile.TemporaryDirectory() as d:
    t0 = time.time()
    ds = make_small(SmallConfig(seed=11, n_dnms=n_dnms))
    paths = dump_dataset(ds, d)
    kid = list(ds.pedigrees)[0]
    bam = paths["bams"][kid]
    print("generated %d DNMs, BAM %.1f MB, sites VCF %d records in %.1f s" % (
        len(ds.dnms), os.path.getsize(bam) / 1e6, len(ds.sites), time.time() - t0), flush=True)
    t = time.time(); _, st = session.load_sites(paths["sites"]); t_sites = time.time() - t
    dnms = [dict(x, bam=bam, cram_ref=None) for x in ds.dnms]
    args = (dnms, list(ds.pedigrees), ds.pedigrees, paths["sites"], 1, 38, False, 1000000, True,
            [0.0, 0.2], [0.8, 1.0], [0.2, 0.8], 20, 10, 5000, 1000000, 3, 1, 151, 2)
    results = {}
    for mode in ("whole", "index"):
        session._READS.clear()
        session._HOSTS.clear()
        if mode == "index":
            t = time.time(); write_bai(bam); print("(BAI written by the Python test writer in %.1f s)" % (time.time() - t))
        stats = []
        orig = io_native.read_bam_regions

        def spy(*a, **k):
            tt = time.time()
            r = orig(*a, **k)
            stats.append((time.time() - tt, r.io_stats))
            return r
        io_native.read_bam_regions = spy
        t = time.time()
        if mode == "whole":
            rt = session.load_reads(bam)
        t_dec = time.time() - t
        for rep in range(2):
            stats.clear()
            dn = [dict(x) for x in dnms]
            t = time.time()
            recs = phase_snvs(dn, *args[1:])
            t_phase = time.time() - t
        io_native.read_bam_regions = orig
        out = os.path.join(d, "out_%s.bed" % mode)
        t = time.time(); write_bed_output(recs, False, False, out, 10); t_bed = time.time() - t
        results[mode] = open(out).read()
        if mode == "whole":
            print("whole : decode sites %.3f s | decode BAM %.3f s (%d records) | phase_snvs %.3f s (%d records out, 2nd call) | BED %.3f s"
                  % (t_sites, t_dec, rt.n_segs, t_phase, len(recs), t_bed))
            print("        end to end %.0f DNMs/s from files" % (len(dnms) / (t_sites + t_dec + t_phase + t_bed)))
        else:
            reg = [s for s in stats if s[1]["records_kept"]]
            print("index : phase_snvs incl. region decode %.3f s (2nd call; region decode %.3f s: %d of the file's blocks inflated, "
                  "%d records walked, %d kept) | BED %.3f s" % (t_phase, sum(s[0] for s in reg), sum(s[1]["blocks_inflated"] for s in reg),
                                                               sum(s[1]["records_walked"] for s in reg), sum(s[1]["records_kept"] for s in reg), t_bed))
            print("        end to end %.0f DNMs/s from files" % (len(dnms) / (t_sites + t_phase + t_bed)))
    assert results["whole"] == results["index"], "the two ways of reading the BAM gave different BED text"
    print("BED text identical for both")
