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


def get_bam_names(bam_dir, bam_pairs, cram_ref):
    bam_dict = {}
    cram_found = False
    if bam_dir is not None:
        for ext in ("*.bam", "*.cram"):
            for path in glob(os.path.join(bam_dir, ext)):
                cram_found = cram_found or ext == "*.cram"
                bam_dict.setdefault(os.path.splitext(os.path.basename(path))[0], set()).add(path)
    if bam_pairs is not None:
        for sample_id, bam in bam_pairs:
            if not os.path.exists(bam) or not os.path.isfile(bam):
                sys.exit("invalid filename " + bam)
            bam_dict[sample_id] = {bam}  # one file per id: overrides the directory scan
            if bam[-4:] == "cram":
                cram_found = True
    if cram_found:
        if cram_ref is None:
            sys.exit("Missing reference file for CRAM")
        elif not os.path.isfile(cram_ref):
            sys.exit("Reference file is not valid")
    return bam_dict


def parse_ped(ped, kids):
    labels = ["kid", "dad", "mom", "sex"]
    kid_entries = {}
    missing_parents = []
    with open(ped, "r") as pedfile:
        for line in pedfile:
            fields = line.strip().split()
            if fields[1] in kids:
                if fields[2] == "0" or fields[3] == "0":
                    if not QUIET_MODE:
                        print("Parent of sample {} missing from pedigree file, will be skipped".format(fields[1]),
                              file=sys.stderr)
                    missing_parents.append(fields[1])
                    continue
                kid_entries[fields[1]] = dict(zip(labels, fields[1:5]))
    for sample in kids:
        if (sample not in kid_entries) and (sample not in missing_parents) and not QUIET_MODE:
            print("{} missing from pedigree file, will be skipped".format(sample), file=sys.stderr)
    return kid_entries


UOPS_HEADER = ('##FORMAT=<ID=UOPS,Number=1,Type=Float,Description="Count of pieces of evidence supporting the '
               'unfazed-identified origin parent or `-1` if missing">')
UET_HEADER = ('##FORMAT=<ID=UET,Number=1,Type=Float,Description="Unfazed evidence type: `0` (readbacked), '
              '`1` (allele-balance, for CNVs only), `2` (both), `3` (ambiguous readbacked), '
              '`4` (ambiguous allele-balance), `5` (ambiguous both), '
              '`6` (auto-phased sex-chromosome variant in male), or `-1` (missing)">')


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
    if in_vcf_name.endswith("bcf") or _is_bcf(in_vcf_name):
        sys.exit("annotated VCF output needs a text VCF as --dnms (BCF input carries no text records): "
                 "rerun with `--output-type bed`")
    """reference unfazed.py:337-441: GT of a phased sample becomes 1|0 (paternal) / 0|1 (maternal),
    every sample gets UOPS and UET appended."""
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


def unfazed(args):
    global QUIET_MODE
    bam_names_dict = get_bam_names(args.bam_dir, args.bam_pairs, args.reference)
    snvs, svs = [], []
    if args.dnms.endswith(".bed"):
        reader, input_type = read_vars_bed, "bed"
    elif args.dnms.endswith(".bed.gz"):
        reader, input_type = read_vars_bedzip, "bed"
    elif True in [args.dnms.endswith(t) for t in VCF_TYPES]:
        reader, input_type = read_vars_vcf, "vcf"
    else:
        sys.exit("dnms file type is unrecognized. Must be bed, bed.gz, vcf, vcf.gz, or bcf")
    QUIET_MODE = args.quiet
    output_type = args.output_type if args.output_type is not None else input_type
    if output_type == "vcf" and input_type != "vcf":
        print("Invalid option: --output-type is vcf, but input is not a vcf type. "
              + "Rerun with `--output-type bed` or input dnms as one of the following:", ", ".join(VCF_TYPES),
              file=sys.stderr)
        sys.exit(1)
    kids = set()
    missing_samples, duplicated_samples = set(), set()
    for var_fields in reader(args.dnms):
        sample = var_fields["kid"]
        if sample not in bam_names_dict:
            if sample not in missing_samples:
                if not QUIET_MODE:
                    print("missing alignment file for", sample, file=sys.stderr)
                missing_samples.add(sample)
            continue
        elif len(bam_names_dict[sample]) != 1:
            if sample not in duplicated_samples:
                if not QUIET_MODE:
                    print("multiple alignment files for", sample + ".",
                          "Please specify correct alignment file using --bam-pairs", file=sys.stderr)
                duplicated_samples.add(sample)
            continue
        kids.add(sample)
        var_fields["bam"] = list(bam_names_dict[sample])[0]
        var_fields["cram_ref"] = args.reference
        if var_fields["vartype"].upper() in SV_TYPES:
            svs.append(var_fields)
        elif var_fields["vartype"].upper() in SNV_TYPES:
            snvs.append(var_fields)
    pedigrees = parse_ped(args.ped, kids)
    kids = list(pedigrees.keys())
    snvs = [v for v in snvs if v["kid"] in kids]
    svs = [v for v in svs if v["kid"] in kids]
    phased_svs, phased_snvs = {}, {}
    if (len(snvs) + len(svs)) == 0:
        sys.exit("No phaseable variants")
    common = (args.threads, args.build, args.no_extended, args.multiread_proc_min, args.quiet, args.ab_homref,
              args.ab_homalt, args.ab_het, args.min_gt_qual, args.min_depth, args.search_dist,
              args.insert_size_max_sample, args.stdevs, args.min_map_qual, args.readlen, args.split_error_margin)
    if len(svs) > 0:
        phased_svs = phase_svs(svs, kids, pedigrees, args.sites, *common, evidence_min_ratio=args.evidence_min_ratio,
                               allele_balance_only=getattr(args, "sv_allele_balance_only", False))
    if len(snvs) > 0:
        phased_snvs = phase_snvs(snvs, kids, pedigrees, args.sites, *common, evidence_min_ratio=args.evidence_min_ratio)
    all_phased = phased_snvs
    all_phased.update(phased_svs)
    if output_type == "vcf":
        write_vcf_output(args.dnms, all_phased, args.include_ambiguous, args.verbose, args.outfile, args.evidence_min_ratio)
    elif output_type == "bed":
        write_bed_output(all_phased, args.include_ambiguous, args.verbose, args.outfile, args.evidence_min_ratio)
