"""Final vote -> call, and the BED text surface.

Host-side mirror of reference unfazed/unfazed.py: summarize_autophased (:162-187),
summarize_record (:190-334) and write_bed_output (:444-515).  The read-backed
integer decision is also computed on the device (uz_phase `origin` / `evidence`);
this module produces the merged record (incl. the CNV allele-balance evidence)
and the exact output text, for any backend.
"""
from __future__ import annotations

import sys
from typing import Dict, List, Optional

BED_HEADER = [
    "#chrom", "start", "end", "vartype", "kid", "origin_parent", "other_parent", "evidence_count", "evidence_types",
]
BED_VERBOSE = ["origin_parent_sites", "origin_parent_reads", "other_parent_sites", "other_parent_reads"]


def summarize_autophased(rec: dict, verbose: bool) -> dict:
    """reference unfazed.py:162-187: chrY -> dad, anything else (chrX) -> mom."""
    chrom = rec["region"]["chrom"]
    on_y = chrom.lower().strip("chr") == "y"
    out = {
        "chrom": chrom,
        "start": int(rec["region"]["start"]),
        "end": int(rec["region"]["end"]),
        "vartype": rec["vartype"],
        "kid": rec["kid"],
        "origin_parent": rec["dad"] if on_y else rec["mom"],
        "other_parent": rec["mom"] if on_y else rec["dad"],
        "evidence_count": 1,
        "evidence_types": ["SEX-CHROM"],
    }
    if verbose:
        for k in BED_VERBOSE:
            out[k] = "NA"
    return out


def _join(items) -> str:
    return ",".join(items) if len(items) > 0 else "-"


def summarize_record(rec: dict, include_ambiguous: bool, verbose: bool, evidence_min_ratio: int) -> Optional[dict]:
    """reference unfazed.py:190-334."""
    if rec["evidence_type"] == "SEX-CHROM":
        return summarize_autophased(rec, verbose)
    dad, mom = rec["dad"], rec["mom"]
    n_dad, n_mom = len(rec["dad_reads"]), len(rec["mom_reads"])
    origin = other = None
    o_sites: List[str] = []
    o_reads: List[str] = []
    x_sites: List[str] = []
    x_reads: List[str] = []
    count = 0
    types: List[str] = []
    ambig = False

    # read-backed evidence (:206-234)
    if n_dad > 0 and n_dad >= evidence_min_ratio * n_mom:
        origin, other, count = dad, mom, len(rec["dad_sites"])
        o_sites += rec["dad_sites"]; o_reads += rec["dad_reads"]
        x_sites += rec["mom_sites"]; x_reads += rec["mom_reads"]
        types.append("READBACKED")
    elif n_mom > 0 and n_mom >= evidence_min_ratio * n_dad:
        origin, other, count = mom, dad, len(rec["mom_sites"])
        o_sites += rec["mom_sites"]; o_reads += rec["mom_reads"]
        x_sites += rec["dad_sites"]; x_reads += rec["dad_reads"]
        types.append("READBACKED")
    elif n_dad > 0 and n_mom > 0:
        origin, count = dad + "|" + mom, n_dad + n_mom
        o_sites += rec["dad_sites"]; o_reads += rec["dad_reads"]
        x_sites += rec["mom_sites"]; x_reads += rec["mom_reads"]
        types.append("AMBIGUOUS_READBACKED")
        ambig = True

    # allele-balance evidence (:236-298)
    c_dad, c_mom = len(rec["cnv_dad_sites"]), len(rec["cnv_mom_sites"])
    if c_dad > 0 and c_dad >= evidence_min_ratio * c_mom:
        if origin == mom and "READBACKED" not in types:
            origin = None
            count += c_dad + c_mom
            o_sites += rec["cnv_dad_sites"]
            x_sites = rec["cnv_mom_sites"]
            types = ["AMBIGUOUS_BOTH"]
            ambig = True
        else:
            origin, other, count = dad, mom, c_dad
            o_sites += rec["cnv_dad_sites"]; o_reads += rec["dad_reads"]
            x_sites += rec["mom_sites"]; x_reads += rec["mom_reads"]
            if "AMBIGUOUS_READBACKED" in types:
                types.remove("AMBIGUOUS_READBACKED")
                ambig = False
            types.append("ALLELE-BALANCE")
    elif c_mom > 0 and c_mom >= evidence_min_ratio * c_dad:
        if origin == dad and "READBACKED" not in types:
            origin = None
            count += c_dad + c_mom
            o_sites += rec["cnv_dad_sites"]
            x_sites += rec["cnv_mom_sites"]
            types = ["AMBIGUOUS_BOTH"]
            ambig = True
        else:
            origin, other, count = mom, dad, c_mom
            o_sites += rec["cnv_mom_sites"]; o_reads += rec["mom_reads"]
            x_sites += rec["dad_sites"]; x_reads += rec["dad_reads"]
            if "AMBIGUOUS_READBACKED" in types:
                types.remove("AMBIGUOUS_READBACKED")  # `ambig` stays set on this branch (:286-287)
            types.append("ALLELE-BALANCE")
    elif (c_dad + c_mom) > 0 and "READBACKED" not in types:
        origin = None
        count += c_dad + c_mom
        o_sites += rec["cnv_dad_sites"]
        x_sites = rec["cnv_mom_sites"]
        types.append("AMBIGUOUS_ALLELE-BALANCE")
        ambig = True

    if (origin is None or ambig) and not include_ambiguous:
        return None
    out = {
        "chrom": rec["region"]["chrom"],
        "start": int(rec["region"]["start"]),
        "end": int(rec["region"]["end"]),
        "vartype": rec["vartype"],
        "kid": rec["kid"],
        "origin_parent": origin,
        "other_parent": other,
        "evidence_count": count,
        "evidence_types": types,
    }
    if verbose:
        out["origin_parent_sites"] = _join(sorted(o_sites))  # sorted as strings (:302-303)
        out["origin_parent_reads"] = _join(o_reads)
        out["other_parent_sites"] = _join(sorted(x_sites))
        out["other_parent_reads"] = _join(x_reads)
    return out


def bed_lines(records: Dict[str, dict], include_ambiguous: bool, verbose: bool, evidence_min_ratio: int) -> List[str]:
    """Header + body lines of reference write_bed_output (unfazed.py:444-515)."""
    cols = list(BED_HEADER) + (BED_VERBOSE if verbose else [])
    rows = []
    for key in records:
        s = summarize_record(records[key], include_ambiguous, verbose, evidence_min_ratio)
        if s is not None:
            rows.append(s)
    rows.sort(key=lambda x: (x["chrom"], x["start"], x["end"]))  # chrom compared as a string (:497-499)
    lines = ["\t".join(cols)]
    for s in rows:
        s = dict(s)
        s["evidence_types"] = ",".join(s["evidence_types"])
        lines.append("\t".join(str(s[c.lstrip("#")]) for c in cols))
    return lines


def write_bed_output(records, include_ambiguous, verbose, outfile, evidence_min_ratio):
    lines = bed_lines(records, include_ambiguous, verbose, evidence_min_ratio)
    if outfile == "/dev/stdout":
        for ln in lines:
            print(ln)
    else:
        with open(outfile, "w") as fh:
            for ln in lines:
                print(ln, file=fh)
