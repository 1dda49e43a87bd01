"""End-to-end timing from files on disk to the BED text, stage by stage (GPU box):
file decode (native decoders) -> phase_snvs (host logic + upload + kernels + records) -> BED writer.
usage: python scripts/e2e_time.py [n_dnms]"""
import io
import os
import sys
import tempfile
import time
import contextlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synth.small import SmallConfig, make_small  # noqa: E402
from tests.filesio import dump_dataset  # noqa: E402
from unfazed_amd import session  # noqa: E402
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
    t = time.time(); rt = session.load_reads(bam); t_reads = time.time() - t
    dnms = [dict(x, bam=bam, cram_ref=None) for x in ds.dnms]
    args = (dnms, list(ds.pedigrees), ds.pedigrees, paths["sites"], 1, 38, False, 1000000, True,
            [0.0, 0.2], [0.8, 1.0], [0.2, 0.8], 20, 10, 5000, 1000000, 3, 1, 151, 2)
    for rep in range(2):
        dn = [dict(x) for x in dnms]
        t = time.time()
        recs = phase_snvs(dn, *args[1:])
        t_phase = time.time() - t
    out = os.path.join(d, "out.bed")
    t = time.time(); write_bed_output(recs, False, False, out, 10); t_bed = time.time() - t
    print("decode sites %.3f s | decode BAM %.3f s (%d records, %s) | phase_snvs %.3f s (%d records out, 2nd call) | BED %.3f s" % (
        t_sites, t_reads, rt.n_segs, getattr(rt, "decode_seconds", None), t_phase, len(recs), t_bed))
    print("end to end: %.0f DNMs/s from files; phase_snvs alone %.0f DNMs/s" % (
        len(dnms) / (t_sites + t_reads + t_phase + t_bed), len(dnms) / t_phase))
