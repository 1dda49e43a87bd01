"""Files -> BED text through the PRODUCT path (session + hostpath + HipEngine) on files of real size (GPU box).

Writes, with the generator's native writer (synth/uzfiles.cpp), a coordinate-sorted BAM + BAI holding the 30x pile-ups of N DNMs
(default 72 000: a file of 5.1 GB at deflate level 6) and the whole 20 M-site table as a BGZF VCF + TBI; then runs
`phase_snvs(...)` -- the drop-in call of the reference's seam -- on the first M of those DNMs (default 20 000) and writes the BED.
Per batch the session decodes only the windows of the VCF (tabix) and only the BGZF blocks of the BAM its fetches reach (BAI,
uz_bam_stage_*: straight into the link form).  Printed: stage times, the file's blocks against the blocks inflated, DNMs/s from
files, calls against the simulated truth.
    python scripts/e2e_time.py [N_in_file=72000] [M_phased=20000]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from synth import bigsynth  # noqa: E402
from synth.sites_np import make_clusters, make_sites, place_dnms_full  # noqa: E402
from unfazed_amd import io_native, session  # noqa: E402
from unfazed_amd.snv_phaser import phase_snvs  # noqa: E402
from unfazed_amd.unfazed import write_bed_output  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 72000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
M = min(M, N)
base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (64 << 30) else None
d = tempfile.mkdtemp(prefix="uze2e_", dir=base)
try:
    t0 = time.time()
    sc = make_sites(20_000_000, seed=202)
    dn = place_dnms_full(sc, 100000, seed=201)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=203)
    cfg.n_clusters = cl.n
    c_hi = cl.of_dnm(N - 1) + 1
    n_in_file = int(cl.d0[c_hi - 1] + cl.nd[c_hi - 1])
    print("generated the site table and the DNM list in %.1f s" % (time.time() - t0), flush=True)
    bam, vcf = os.path.join(d, "kid.bam"), os.path.join(d, "sites.vcf.gz")
    t = time.time()
    sb = bigsynth.write_bam(bam, cfg, sc, dn, cl, 0, c_hi, level=6)
    print("BAM + BAI: %d records of %d DNMs, %.2f GB on disk (%.1f GB inflated, %d BGZF blocks), written in %.1f s" % (
        sb["records"], n_in_file, sb["file_bytes"] / 1e9, sb["raw_bytes"] / 1e9, sb["blocks"], time.time() - t), flush=True)
    t = time.time()
    sv = bigsynth.write_vcf(vcf, sc, level=6)
    print("VCF + TBI: %d records, %.2f GB on disk (%d blocks), written in %.1f s" % (sv["records"], sv["file_bytes"] / 1e9, sv["blocks"], time.time() - t), flush=True)
    ped = {"kid": {"kid": "kid", "dad": "dad", "mom": "mom", "sex": "2"}}
    dnms = [dict(chrom=sc.contig_names[int(c)], start=int(s), end=int(e), kid="kid", vartype="POINT", bam=bam, cram_ref=None)
            for c, s, e in zip(dn.contig[:M], dn.start[:M], dn.end[:M])]
    args = (["kid"], ped, vcf, 2, "38", False, 10 ** 9, True, [0.0, 0.2], [0.8, 1.0], [0.2, 0.8], 20, 10, 5000, 1000000, 3, 1, 151, 5)
    # spy on the two decoders the session calls
    stats = {"bam": [], "vcf": []}
    orig_sel, orig_vcf = io_native.BamSource.select, io_native.read_vcf_table_regions

    def spy_sel(self, *a, **k):
        tt = time.time()
        r = orig_sel(self, *a, **k)
        stats["bam"].append((time.time() - tt, r.io_stats, r.timing))
        return r

    def spy_vcf(*a, **k):
        tt = time.time()
        r = orig_vcf(*a, **k)
        stats["vcf"].append((time.time() - tt, r.io_stats))
        return r
    orig_kept = io_native.BamSource.select_kept

    def spy_kept(self, *a, **k):  # the device-walk route (include/uz_bamwalk.h): the blocks are inflated and walked on the device
        tt = time.time()
        r = orig_kept(self, *a, **k)
        io = dict(r.io_stats)
        io["blocks_inflated"] = r.plan["n_blocks"]
        io["walked_on_the_device"] = 1
        stats["bam"].append((time.time() - tt, io, r.timing))
        return r
    io_native.BamSource.select = spy_sel
    io_native.BamSource.select_kept = spy_kept
    io_native.read_vcf_table_regions = spy_vcf
    for rep in range(2):
        for v in stats.values():
            v.clear()
        session._HOSTS.clear()
        for k in [k for k in session._SITES if "@" in k]:
            del session._SITES[k]
        batch = [dict(x) for x in dnms]
        t = time.time()
        recs = phase_snvs(batch, *args)
        t_phase = time.time() - t
    out = os.path.join(d, "out.bed")
    t = time.time()
    write_bed_output(recs, False, False, out, 10)
    t_bed = time.time() - t
    bam_s = sum(s[0] for s in stats["bam"])
    blocks = sum(s[1]["blocks_inflated"] for s in stats["bam"])
    walked = sum(s[1]["records_walked"] for s in stats["bam"])
    kept = sum(s[1]["records_kept"] for s in stats["bam"])
    vcf_s = sum(s[0] for s in stats["vcf"])
    vcf_blocks = sum(s[1][1] for s in stats["vcf"])
    vcf_kept = sum(s[1][3] for s in stats["vcf"])
    print("phase_snvs on %d DNMs from the files (2nd call): %.2f s | BAM stage %.2f s: %d of the file's %d blocks inflated (%.1f %%), %d records walked, "
          "%d kept | VCF windows %.2f s: %d of %d blocks, %d records | BED %.2f s" % (
              M, t_phase, bam_s, blocks, sb["blocks"], 100.0 * blocks / sb["blocks"], walked, kept, vcf_s, vcf_blocks, sv["blocks"], vcf_kept, t_bed), flush=True)
    print("end to end %.0f DNMs/s from files to BED text (%d records); inflate %s, %d host threads (cgroup quota %s CPUs)" % (
        M / (t_phase + t_bed), len(recs), ("the device (records walked there too), " if sum(s[1].get("walked_on_the_device", 0) for s in stats["bam"]) else "the device, " if os.environ.get("UZ_INFLATE", "device") == "device" and sum(s[1].get("blocks_from_the_device", 0) for s in stats["bam"]) else "")
        + io_native.inflate_backend(), io_native.default_threads(), io_native.cpu_quota() or "no"), flush=True)
    # calls against the simulated truth
    truth = {"%s_%d_%d_kid_POINT" % (sc.contig_names[int(c)], int(s), int(e)): ("dad" if o == 0 else "mom") for c, s, e, o in zip(dn.contig[:M], dn.start[:M], dn.end[:M], dn.origin[:M])}
    called = right = 0
    for line in open(out):
        f = line.rstrip("\n").split("\t")
        if f[0].startswith("#") or len(f) < 6:
            continue
        key = "%s_%s_%s_kid_POINT" % (f[0], f[1], f[2])
        if f[5] in ("dad", "mom") and key in truth:
            called += 1
            right += truth[key] == f[5]
    print("calls: %d of %d DNMs phased to one parent, %d equal to the simulated origin" % (called, M, right))
    assert called > M // 5 and right >= called * 0.98
finally:
    shutil.rmtree(d, ignore_errors=True)
