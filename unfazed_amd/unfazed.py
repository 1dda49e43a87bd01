"""Driver and I/O surface -- host glue mirroring reference unfazed/unfazed.py
(DNM readers :18-90, get_bam_names :93-126, parse_ped :129-159, write_vcf_output :337-441,
unfazed :518-667).  Tiny data, no acceleration target; kept so that the command line and
the BED / annotated-VCF outputs drop in."""
from __future__ import annotations

import gzip
import os
import sys

import numpy as np
from glob import glob

from . import __reference_version__
from .hostpath import SNV_TYPES, SV_TYPES
from .io_vcf import read_vcf
from .model import HET, HOM_ALT
from .snv_phaser import phase_snvs
from .summarize import summarize_record, write_bed_output
from .sv_phaser import phase_svs

VCF_TYPES = ["vcf", "vcf.gz", "bcf"]  # reference utils.py:7
LABELS = ["chrom", "start", "end", "kid", "vartype"]
QUIET_MODE = False


def _bed_rows(lines):
    for line in lines:
        if line[0] == "#":
            continue
        fields = line.strip().split()
        if not len(fields) == 5:
            sys.exit("dnms bed file must contain the following columns exactly: " + ", ".join(LABELS))
        vartype = fields[4]
        if vartype not in SV_TYPES:
            vartype = SNV_TYPES[0]  # anything else becomes POINT (quirk Q20)
        yield {"chrom": fields[0], "start": int(fields[1]), "end": int(fields[2]), "kid": fields[3],
               "vartype": vartype, "bam": ""}


def read_vars_bed(bedname):
    with open(bedname, "r") as fh:
        yield from _bed_rows(fh)


def read_vars_bedzip(bedzipname):
    # the reference opens in binary mode and compares bytes to str (:44-48), which fails on
    # Python 3; a text-mode reader is the evident intent.
    with gzip.open(bedzipname, "rt") as fh:
        yield from _bed_rows(fh)


def _is_bcf(name):
    import gzip
    try:
        with gzip.open(name, "rb") as fh:
            return fh.read(4) == b"BCF\x02"
    except OSError:
        return False


def read_vars_vcf(vcfname):
    if vcfname.endswith("bcf") or _is_bcf(vcfname):
        # binary form: through the native decoder (io_native); same fields as the text path below
        from .io_native import read_vcf_table
        t = read_vcf_table(vcfname)
        chrom_of = np.repeat(np.arange(len(t.contigs)), np.diff(t.contig_off))
        for j in range(t.n_sites):
            vartype = t.info(j, "SVTYPE")
            if vartype is None:
                vartype = SNV_TYPES[0]
            for i in range(len(t.samples)):
                if int(t.gt[i, j]) in [HET, HOM_ALT]:
                    yield {"chrom": t.contigs[int(chrom_of[j])], "start": int(t.pos[j]), "end": int(t.end[j]),
                           "kid": t.samples[i], "vartype": vartype, "bam": ""}
        return
    samples, records, _ = read_vcf(vcfname)
    for r in records:
        vartype = r.info.get("SVTYPE")
        if vartype is None:
            vartype = SNV_TYPES[0]
        for i, gt in enumerate(r.gt_types):
            if gt in [HET, HOM_ALT]:
                yield {"chrom": r.chrom, "start": r.start, "end": r.end, "kid": samples[i], "vartype": vartype, "bam": ""}


class AlignmentFiles:
    """sample id -> alignment file(s): the directory scan (`{sample_id}.bam` / `.cram`) overlaid by explicit
    `id:path` pairs, which win.  Same outcomes and messages as reference unfazed.py:93-126."""

    def __init__(self):
        self.by_sample = {}
        self.saw_cram = False

    def scan_dir(self, directory):
        for path in sorted(glob(os.path.join(directory, "*.bam"))) + sorted(glob(os.path.join(directory, "*.cram"))):
            stem, ext = os.path.splitext(os.path.basename(path))
            self.by_sample.setdefault(stem, set()).add(path)
            self.saw_cram |= ext == ".cram"

    def pin(self, sample_id, path):
        if not os.path.isfile(path):
            sys.exit("invalid filename " + path)
        self.by_sample[sample_id] = {path}
        self.saw_cram |= path.endswith("cram")

    def check_cram_reference(self, cram_ref):
        if not self.saw_cram:
            return
        if cram_ref is None:
            sys.exit("Missing reference file for CRAM")
        if not os.path.isfile(cram_ref):
            sys.exit("Reference file is not valid")

    def lookup(self, sample):
        """-> (path, None) or (None, why) with why in {"missing", "multiple"}"""
        files = self.by_sample.get(sample)
        if files is None:
            return None, "missing"
        if len(files) != 1:
            return None, "multiple"
        return next(iter(files)), None


def get_bam_names(bam_dir, bam_pairs, cram_ref):
    files = AlignmentFiles()
    if bam_dir is not None:
        files.scan_dir(bam_dir)
    for sample_id, path in (bam_pairs or []):
        files.pin(sample_id, path)
    files.check_cram_reference(cram_ref)
    return files.by_sample


def parse_ped(ped, kids):
    """kid id -> {kid, dad, mom, sex} (strings) for the kids that have DNMs; kids with a parent coded `0`
    and kids absent from the file are reported and skipped (reference unfazed.py:129-159)."""
    wanted = set(kids)
    entries, parentless = {}, set()
    with open(ped, "r") as fh:
        rows = [ln.split() for ln in fh]
    for row in rows:
        sample = row[1]
        if sample not in wanted:
            continue
        dad, mom = row[2], row[3]
        if "0" in (dad, mom):
            _say("Parent of sample {} missing from pedigree file, will be skipped".format(sample))
            parentless.add(sample)
        else:
            entries[sample] = {"kid": sample, "dad": dad, "mom": mom, "sex": row[4]}
    for sample in kids:
        if sample not in entries and sample not in parentless:
            _say("{} missing from pedigree file, will be skipped".format(sample))
    return entries


def _say(*words):
    if not QUIET_MODE:
        print(*words, file=sys.stderr)


UOPS_HEADER = ('##FORMAT=<ID=UOPS,Number=1,Type=Float,Description="Count of pieces of evidence supporting the '
               'unfazed-identified origin parent or `-1` if missing">')
UET_HEADER = ('##FORMAT=<ID=UET,Number=1,Type=Float,Description="Unfazed evidence type: `0` (readbacked), '
              '`1` (allele-balance, for CNVs only), `2` (both), `3` (ambiguous readbacked), '
              '`4` (ambiguous allele-balance), `5` (ambiguous both), '
              '`6` (auto-phased sex-chromosome variant in male), or `-1` (missing)">')


BCF_OUTPUT_MESSAGE = ("annotated VCF output needs a text VCF as --dnms (BCF input carries no text records): "
                      "rerun with `--output-type bed`")


def uet_code(evidence_types):
    """reference unfazed.py:415-433"""
    if "AMBIGUOUS_READBACKED" in evidence_types:
        return 3
    if "AMBIGUOUS_ALLELE-BALANCE" in evidence_types:
        return 4
    if "AMBIGUOUS_BOTH" in evidence_types:
        return 5
    if "SEX-CHROM" in evidence_types:
        return 6
    if "READBACKED" in evidence_types and "ALLELE-BALANCE" in evidence_types:
        return 2
    if "READBACKED" in evidence_types:
        return 0
    if "ALLELE-BALANCE" in evidence_types:
        return 1
    return -1


def write_vcf_output(in_vcf_name, read_records, include_ambiguous, verbose, outfile, evidence_min_ratio):
    """reference unfazed.py:337-441: GT of a phased sample becomes 1|0 (paternal) / 0|1 (maternal),
    every sample gets UOPS and UET appended."""
    if in_vcf_name.endswith("bcf") or _is_bcf(in_vcf_name):
        sys.exit(BCF_OUTPUT_MESSAGE)
    samples, records, header = read_vcf(in_vcf_name)
    out = []
    out.extend(header[:-1])
    out.append("##unfazed=" + __reference_version__
               + ". Phase info in pipe-separated GT field order -> 1|0 is paternal, 0|1 is maternal")
    out.append(UOPS_HEADER)
    out.append(UET_HEADER)
    out.append(header[-1])
    for r in records:
        f = list(r.raw)
        fmt = f[8].split(":")
        gt_i = fmt.index("GT") if "GT" in fmt else None
        f[8] = f[8] + ":UOPS:UET"
        for i, gt in enumerate(r.gt_types):
            uops, uet = -1, -1
            col = f[9 + i].split(":")
            # VCF lets a sample drop trailing FORMAT fields (`./.`): pad, so UOPS / UET land in their own columns
            col += ["."] * (len(fmt) - len(col))
            if gt in [HET, HOM_ALT]:
                vartype = r.info.get("SVTYPE")
                if vartype is None:
                    vartype = SNV_TYPES[0]
                key = "{}_{}_{}_{}_{}".format(r.chrom, r.start, r.end, samples[i], vartype)
                if key in read_records:
                    s = summarize_record(read_records[key], include_ambiguous, verbose, evidence_min_ratio)
                    if s is not None:
                        if gt_i is not None:
                            if s["origin_parent"] == read_records[key]["dad"]:
                                col[gt_i] = "1|0"
                            elif s["origin_parent"] == read_records[key]["mom"]:
                                col[gt_i] = "0|1"
                        uops = s["evidence_count"]
                        uet = uet_code(s["evidence_types"])
            f[9 + i] = ":".join(col) + ":%g:%g" % (uops, uet)
        out.append("\t".join(f))
    text = "\n".join(out) + "\n"
    if outfile == "/dev/stdout":
        sys.stdout.write(text)
    else:
        with open(outfile, "w") as fh:
            fh.write(text)


def _dnm_reader(path):
    if path.endswith(".bed"):
        return read_vars_bed, "bed"
    if path.endswith(".bed.gz"):
        return read_vars_bedzip, "bed"
    if any(path.endswith(t) for t in VCF_TYPES):
        return read_vars_vcf, "vcf"
    sys.exit("dnms file type is unrecognized. Must be bed, bed.gz, vcf, vcf.gz, or bcf")


def _route_variants(variants, bam_names, cram_ref):
    """Attach the kid's alignment file to every DNM and split the list into point variants and SVs.  A kid
    without exactly one alignment file is reported once and its DNMs are dropped."""
    snvs, svs, kids, reported = [], [], set(), set()
    for var in variants:
        sample = var["kid"]
        files = bam_names.get(sample)
        if files is None or len(files) != 1:
            if sample not in reported:
                reported.add(sample)
                if files is None:
                    _say("missing alignment file for", sample)
                else:
                    _say("multiple alignment files for", sample + ".",
                         "Please specify correct alignment file using --bam-pairs")
            continue
        kids.add(sample)
        var["bam"] = next(iter(files))
        var["cram_ref"] = cram_ref
        kind = var["vartype"].upper()
        if kind in SV_TYPES:
            svs.append(var)
        elif kind in SNV_TYPES:
            snvs.append(var)
    return snvs, svs, kids


def _ranks(args):
    """(rank, world, torch.distributed or None): more than one rank when a launcher (torchrun, or `--gpus N` which starts the ranks
    itself, __main__.spawn_ranks) put RANK / WORLD_SIZE into the environment.  The only communication is the gather of the per-shard
    records (host objects) on rank 0: gloo, whatever the ranks compute on."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, None
    import datetime
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(hours=12))
    return rank, world, dist


def unfazed(args):
    """reference unfazed.py:518-667: read the DNMs, find each kid's alignment file and pedigree entry, phase
    SVs and point variants through the two phasers, write BED or annotated VCF."""
    global QUIET_MODE
    bam_names = get_bam_names(args.bam_dir, args.bam_pairs, args.reference)
    reader, input_type = _dnm_reader(args.dnms)
    QUIET_MODE = args.quiet
    output_type = args.output_type or input_type
    if output_type == "vcf" and input_type != "vcf":
        print("Invalid option: --output-type is vcf, but input is not a vcf type. "
              + "Rerun with `--output-type bed` or input dnms as one of the following:", ", ".join(VCF_TYPES),
              file=sys.stderr)
        sys.exit(1)
    if output_type == "vcf" and (args.dnms.endswith("bcf") or _is_bcf(args.dnms)):
        sys.exit(BCF_OUTPUT_MESSAGE)  # known before any phasing is done
    snvs, svs, kids = _route_variants(reader(args.dnms), bam_names, args.reference)
    pedigrees = parse_ped(args.ped, kids)
    kids = list(pedigrees)
    snvs = [v for v in snvs if v["kid"] in pedigrees]
    svs = [v for v in svs if v["kid"] in pedigrees]
    if not snvs and not svs:
        sys.exit("No phaseable variants")
    thresholds = (args.threads, args.build, args.no_extended, args.multiread_proc_min, args.quiet, args.ab_homref,
                  args.ab_homalt, args.ab_het, args.min_gt_qual, args.min_depth, args.search_dist,
                  args.insert_size_max_sample, args.stdevs, args.min_map_qual, args.readlen, args.split_error_margin)
    # SVs first, as the reference does (:601-646): the order of the per-variant messages on stderr is part of
    # the surface; on a key collision the SV record wins (its final merge, :648-649)
    sv_records, snv_records = {}, {}
    rank, world, dist = _ranks(args)
    if world > 1:
        # one process per GPU, contiguous DNM shards, no collective on the data path: the reference's task pool
        # over DNMs (snv_phaser.py:244-298) as shards; the records are gathered on rank 0, which writes the output.  A batch that
        # takes the many-variant path as a whole (--multiread-proc-min) takes it in every shard.
        from . import session, shard
        session.set_device(int(os.environ.get("LOCAL_RANK", rank)))
        mpm = args.multiread_proc_min

        def shard_thresholds(batch):
            return (args.threads, args.build, args.no_extended, 0 if len(batch) >= mpm else 1 << 60) + thresholds[4:]
        if svs:
            sv_records = shard.phase_sharded(phase_svs, svs, kids, pedigrees, args.sites, *shard_thresholds(svs), rank=rank, world=world, dist=dist,
                                             evidence_min_ratio=args.evidence_min_ratio,
                                             allele_balance_only=getattr(args, "sv_allele_balance_only", False))
        if snvs:
            snv_records = shard.phase_sharded(phase_snvs, snvs, kids, pedigrees, args.sites, *shard_thresholds(snvs), rank=rank, world=world, dist=dist,
                                              evidence_min_ratio=args.evidence_min_ratio)
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return
        sv_records, snv_records = sv_records or {}, snv_records or {}
    else:
        if svs:
            sv_records = phase_svs(svs, kids, pedigrees, args.sites, *thresholds, evidence_min_ratio=args.evidence_min_ratio,
                                   allele_balance_only=getattr(args, "sv_allele_balance_only", False))
        if snvs:
            snv_records = phase_snvs(snvs, kids, pedigrees, args.sites, *thresholds, evidence_min_ratio=args.evidence_min_ratio)
    records = dict(snv_records)
    records.update(sv_records)
    if output_type == "vcf":
        write_vcf_output(args.dnms, records, args.include_ambiguous, args.verbose, args.outfile, args.evidence_min_ratio)
    else:
        write_bed_output(records, args.include_ambiguous, args.verbose, args.outfile, args.evidence_min_ratio)
