"""VCF text decoder (plain / gzip / bgzip) into unfazed_amd.model.SiteRecord.

Replaces what the reference gets from cyvcf2: `Variant.start`,
`end`, `REF`, `ALT`, `INFO.get`, `gt_types`, `gt_ref_depths`, `gt_alt_depths`,
`gt_quals`, `genotypes`.  BCF is not decoded here (text VCF only).

Genotype coding follows cyvcf2 with gts012=False (reference utils.py:2-5): both
alleles called and equal to 0 -> HOM_REF 0; called and different -> HET 1; called,
equal and non-zero -> HOM_ALT 3; missing -> UNKNOWN 2.  Half-missing and haploid calls are
not exercised by any reference test (parity unpinned): a half-missing call counts with its
called allele (0 -> HOM_REF, else HET), a haploid call as homozygous.
Depths come from FORMAT/AD (first ALT), falling back to RO / AO; missing -> -1; GQ -> float,
missing -> -1.
"""
from __future__ import annotations

import gzip
from typing import Iterator, List, Tuple

from .model import GT_UNKNOWN, HET, HOM_ALT, HOM_REF, SiteRecord


def _open_text(path: str):
    with open(path, "rb") as fh:
        magic = fh.read(2)
    if magic == b"\x1f\x8b":
        return gzip.open(path, "rt")
    return open(path, "r")


def _parse_gt(gt: str):
    if gt in (".", "./.", ".|."):
        return GT_UNKNOWN, [-1, -1, False]
    phased = "|" in gt
    parts = gt.replace("|", "/").split("/")
    alleles = [(-1 if p == "." else int(p)) for p in parts]
    if len(alleles) == 1:
        a = alleles[0]
        code = GT_UNKNOWN if a < 0 else (HOM_REF if a == 0 else HOM_ALT)
        return code, [a, phased]
    a, b = alleles[0], alleles[1]
    if a < 0 and b < 0:
        code = GT_UNKNOWN
    elif a < 0 or b < 0:
        c = a if b < 0 else b
        code = HOM_REF if c == 0 else HET
    elif a != b:
        code = HET
    elif a == 0:
        code = HOM_REF
    else:
        code = HOM_ALT
    return code, [a, b, phased]


def _num(x: str, cast, missing=-1):
    if x in (".", ""):
        return missing
    try:
        return cast(x)
    except ValueError:
        return missing


def read_vcf(path: str) -> Tuple[List[str], List[SiteRecord], List[str]]:
    """-> (samples, records in file order, header lines incl. the #CHROM line)"""
    samples: List[str] = []
    header: List[str] = []
    records: List[SiteRecord] = []
    with _open_text(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.startswith("##"):
                header.append(line)
                continue
            if line.startswith("#"):
                header.append(line)
                samples = line.split("\t")[9:]
                continue
            f = line.split("\t")
            chrom, pos, ref = f[0], int(f[1]), f[3]
            alts = [] if f[4] == "." else f[4].split(",")
            info = {}
            if len(f) > 7 and f[7] != ".":
                for kv in f[7].split(";"):
                    if "=" in kv:
                        k, v = kv.split("=", 1)
                        info[k] = v
                    else:
                        info[kv] = True
            start = pos - 1
            end = start + len(ref)
            if "END" in info:
                try:
                    end = int(info["END"])
                except ValueError:
                    pass
            ns = len(samples)
            gts, rds, ads, gqs, genos = [GT_UNKNOWN] * ns, [-1] * ns, [-1] * ns, [-1.0] * ns, []
            fmt = f[8].split(":") if len(f) > 8 else []
            idx = {k: i for i, k in enumerate(fmt)}
            for s in range(ns):
                col = f[9 + s].split(":") if len(f) > 9 + s else ["."]
                code, g = _parse_gt(col[idx["GT"]]) if "GT" in idx and idx["GT"] < len(col) else (GT_UNKNOWN, [-1, -1, False])
                gts[s] = code
                genos.append(g)
                if "AD" in idx and idx["AD"] < len(col) and col[idx["AD"]] != ".":
                    ad = col[idx["AD"]].split(",")
                    rds[s] = _num(ad[0], int)
                    ads[s] = _num(ad[1], int) if len(ad) > 1 else -1
                elif "RO" in idx and "AO" in idx and idx["RO"] < len(col) and idx["AO"] < len(col):
                    rds[s] = _num(col[idx["RO"]], int)
                    ads[s] = _num(col[idx["AO"]].split(",")[0], int)
                if "GQ" in idx and idx["GQ"] < len(col):
                    gqs[s] = float(_num(col[idx["GQ"]], float, -1.0))
            records.append(SiteRecord(chrom, start, ref, alts, gts, rds, ads, gqs, end=end, info=info,
                                      genotypes=genos, raw=f))
    return samples, records, header
