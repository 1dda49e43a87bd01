"""Known answers for the input readers on the reference's OWN small test files (copied as data into
tests/golden/refdata/): the variants its read_vars_bed / read_vars_vcf yield and the pedigree entries and
messages of its parse_ped (unfazed/unfazed.py:18-162).  SURVEY.md 8(d) config 1: the bundled BAM and sites
VCF are missing from the snapshot, so these files pin the plumbing only.

Run in the authoring container (imports /root/reference through tests/refshim):
    python tests/golden/make_refdata_golden.py
read_vars_bed and parse_ped are plain Python in the reference and read the files themselves;
read_vars_vcf goes through the cyvcf2 stand-in, which is fed the records our text decoder produced, so for the
VCFs only the reference's iteration / typing logic is pinned, not cyvcf2's parsing."""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refrun  # noqa: E402
from unfazed_amd.io_vcf import read_vcf  # noqa: E402

D = os.path.join(HERE, "refdata")


def main():
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    out = {"bed": {}, "vcf": {}, "ped": {}}
    for name in ("trio_hets_snvs_chr22.bed", "trio_hets_svs_chr22.bed"):
        out["bed"][name] = list(uz.read_vars_bed(os.path.join(D, name)))
    for name in ("trio_hets_snvs_chr22.vcf.gz", "trio_hets_svs_chr22.vcf.gz"):
        samples, recs, _ = read_vcf(os.path.join(D, name))
        key = "mem://refdata/" + name
        cyvcf2.register(key, samples, recs)
        out["vcf"][name] = list(uz.read_vars_vcf(key))
    kids_sets = {"kid": ["NA12878"], "kid+unknown": ["NA12878", "nobody"], "parents": ["NA12891", "NA12892"]}
    for ped in ("trio.ped", "trio_missing_dad.ped", "trio_missing_kid.ped"):
        for label, kids in kids_sets.items():
            err = io.StringIO()
            uz.QUIET_MODE = False
            with contextlib.redirect_stderr(err):
                entries = uz.parse_ped(os.path.join(D, ped), set(kids))
            out["ped"]["%s|%s" % (ped, label)] = {"kids": kids, "entries": entries, "stderr": sorted(err.getvalue().splitlines())}
    with open(os.path.join(HERE, "refdata_plumbing.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print({k: {n: len(v) for n, v in out[k].items()} for k in out})


if __name__ == "__main__":
    main()
