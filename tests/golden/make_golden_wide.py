"""Second batch of golden vectors from the REFERENCE itself (/root/reference through tests/refshim), SURVEY.md 8(c):
  G6  wide_snv_*.json.gz      >= 200 DNMs through phase_snvs, extended and --no-extended (+ a tie-order variant, G5)
  G8  wide_cnv.json.gz        >= 200 DEL / DUP through run_cnv_phasing + summarize_record
      wide_sv_*.json.gz       >= 200 SVs through the whole phase_svs (allele balance + collect_reads_sv + merge)
      autophase.json          chrX / chrY DNMs x sex x build x PAR edges through phase_snvs and phase_svs (quirk Q18)
Large outputs are stored in the compact form of tests/helpers.py (read-name lists as count + digest).
Run in the authoring container only:   python tests/golden/make_golden_wide.py [snv] [cnv] [sv] [auto]"""
import contextlib
import copy
import gzip
import io
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import refrun  # noqa: E402
from helpers import compact_dnms, compact_records, reverse_ties  # noqa: E402
from make_golden import dataset_digest, make_cnv_dataset  # noqa: E402
from synth.small import SmallConfig, make_small  # noqa: E402

WIDE_SNV = [
    ("extended", dict(seed=101, n_dnms=220, kids=["kidA", "kidB"], odd_read_prob=0.06, softclip_prob=0.04, indel_prob=0.03,
                      cluster_prob=0.5, base_err=0.008), dict(), False),
    ("no_extended", dict(seed=102, n_dnms=220, kids=["kidA", "kidB"], odd_read_prob=0.06, indel_dnm_frac=0.3, mnp_dnm_frac=0.1),
     dict(no_extended=True), False),
    ("ties_reversed", dict(seed=103, n_dnms=40, base_err=0.02, cluster_prob=1.0, lowq_prob=0.05, site_rate=1 / 200.0), dict(), True),
    ("ties_forward", dict(seed=103, n_dnms=40, base_err=0.02, cluster_prob=1.0, lowq_prob=0.05, site_rate=1 / 200.0), dict(), False),
]


def dump(name, obj):
    with gzip.GzipFile(os.path.join(HERE, name), "wb", mtime=0) as fh:
        fh.write(json.dumps(obj, sort_keys=True).encode())


def gen_snv():
    for name, cfgkw, runkw, rev in WIDE_SNV:
        ds = make_small(SmallConfig(**cfgkw))
        if rev:
            reverse_ties(ds)
        recs, dnms, err, cutoffs = refrun.run_phase_snvs(ds, tag="wide_" + name, **runkw)
        dump("wide_snv_%s.json.gz" % name, dict(config=cfgkw, run=runkw, reverse_ties=rev, digest=dataset_digest(ds),
                                                record_order=list(recs.keys()), records=compact_records(recs),
                                                dnms=compact_dnms(dnms), stderr=err.splitlines(), cutoffs=cutoffs))
        print("wide snv", name, len(ds.dnms), "DNMs", len(recs), "records")


def gen_cnv():
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    ds = make_cnv_dataset(seed=131, n=210)
    vcf, bams = refrun.register(ds, "wide_cnv")
    cases = []
    for ps in (dict(ab_homref=[0.0, 0.2], ab_homalt=[0.8, 1.0], ab_het=[0.2, 0.8], min_gt_qual=20, min_depth=10),
               dict(ab_homref=[0.0, 0.3], ab_homalt=[0.7, 1.0], ab_het=[0.1, 0.9], min_gt_qual=10, min_depth=4)):
        dnms = copy.deepcopy(ds.dnms)
        err = io.StringIO()
        svp.QUIET_MODE = False
        with warnings.catch_warnings(), contextlib.redirect_stderr(err):
            warnings.simplefilter("ignore")
            recs = svp.run_cnv_phasing(dnms, ds.pedigrees, vcf, 1, "38", 1000, ps["ab_homref"], ps["ab_homalt"], ps["ab_het"],
                                       ps["min_gt_qual"], ps["min_depth"])
        summaries = {k: uz.summarize_record(copy.deepcopy(r), True, True, 10) for k, r in recs.items()}
        cases.append(dict(params=ps, records=recs, record_order=list(recs.keys()), stderr=err.getvalue().splitlines(),
                          dnms=compact_dnms(dnms), summaries=summaries))
        print("wide cnv", len(ds.dnms), "events", len(recs), "records")
    dump("wide_cnv.json.gz", dict(seed=131, n=210, digest=dataset_digest(ds), cases=cases))


WIDE_SV = [("a", dict(seed=141, n_svs=72), dict()), ("b", dict(seed=142, n_svs=72), dict(no_extended=True)),
           ("c", dict(seed=143, n_svs=72), dict(min_gt_qual=10, split_error_margin=2, evidence_min_ratio=10))]


def gen_sv():
    from synth.small_sv import SvConfig, make_small_sv
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    for name, cfgkw, runkw in WIDE_SV:
        ds = make_small_sv(SvConfig(**cfgkw))
        kw = {k: v for k, v in runkw.items() if k != "evidence_min_ratio"}
        recs, dnms, err, cutoffs = refrun.run_phase_svs(ds, tag="wide_sv_" + name, **kw)
        summaries = {k: uz.summarize_record(copy.deepcopy(r), True, False, 10) for k, r in recs.items()}
        dump("wide_sv_%s.json.gz" % name, dict(config=cfgkw, run=kw, digest=dataset_digest(ds), record_order=list(recs.keys()),
                                               records=compact_records(recs), dnms=compact_dnms(dnms), stderr=err.splitlines(),
                                               summaries=summaries))
        print("wide sv", name, len(ds.dnms), "SVs", len(recs), "records",
              sum(1 for r in recs.values() if r["dad_reads"] or r["mom_reads"]), "read-backed")


# PAR coordinates as the reference holds them (utils.py:26-43): every edge, one base either side
PAR = {"37": {"x": [(10001, 2781479), (155701383, 156030895)], "y": [(10001, 2781479), (56887903, 57217415)]},
       "38": {"x": [(60001, 2699520), (154931044, 155260560)], "y": [(10001, 2649520), (59034050, 59363566)]}}


def autophase_dataset(prefix):
    """A small trio pair (kidA male, kidB female) on contigs X, Y, 1 with ordinary DNMs, plus DNMs without any data
    around them at every PAR edge of both builds."""
    ds = make_small(SmallConfig(seed=151, n_dnms=9, contigs=["X", "Y", "1"], kids=["kidA", "kidB"], chr_prefix=prefix))
    extra = []
    for chrom in ("x", "y"):
        pos = set()
        for build in ("37", "38"):
            for lo, hi in PAR[build][chrom]:
                pos.update([lo - 1, lo, hi, hi + 1])
        for p in sorted(pos):
            for kid in ("kidA", "kidB"):
                extra.append({"chrom": prefix + chrom.upper(), "start": p, "end": p + 1, "kid": kid, "vartype": "POINT", "bam": "",
                              "cram_ref": None})
    ds.dnms = ds.dnms + extra
    return ds


def gen_auto():
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    runs = []
    for prefix in ("", "chr"):
        for build in ("37", "38", "na"):
            for many in (False, True):
                ds = autophase_dataset(prefix)
                kw = dict(build=build, multithread_proc_min=1 if many else 1000)
                recs, dnms, err, _ = refrun.run_phase_snvs(ds, tag="auto%s%s%d" % (prefix, build, many), **kw)
                summaries = {k: uz.summarize_record(copy.deepcopy(r), True, True, 10) for k, r in recs.items()}
                runs.append(dict(kind="snv", prefix=prefix, run=kw, digest=dataset_digest(ds), record_order=list(recs.keys()),
                                 records=compact_records(recs), dnms=compact_dnms(dnms), stderr=err.splitlines(), summaries=summaries))
                print("auto snv", prefix, build, many, len(recs), "records", sum(r["evidence_type"] == "SEX-CHROM" for r in recs.values()), "autophased")
            # the SV driver: sv_phaser.autophase writes the record but does not return True (quirk Q18)
            ds = autophase_dataset(prefix)
            for i, d in enumerate(ds.dnms):
                d["vartype"] = ["DEL", "DUP", "INV"][i % 3]
                d["end"] = d["start"] + 700 + 50 * (i % 7)
            kw = dict(build=build)
            recs, dnms, err, _ = refrun.run_phase_svs(ds, tag="autosv%s%s" % (prefix, build), **kw)
            summaries = {k: uz.summarize_record(copy.deepcopy(r), True, True, 10) for k, r in recs.items()}
            runs.append(dict(kind="sv", prefix=prefix, run=kw, digest=dataset_digest(ds), record_order=list(recs.keys()),
                             records=compact_records(recs), dnms=compact_dnms(dnms), stderr=err.splitlines(), summaries=summaries))
            print("auto sv", prefix, build, len(recs), "records", sum(r["evidence_type"].startswith("SEX-CHROM") for r in recs.values()), "autophased")
    with open(os.path.join(HERE, "autophase.json"), "w") as fh:
        json.dump(dict(runs=runs), fh, sort_keys=True)


if __name__ == "__main__":
    assert refrun.available(), "/root/reference is required to generate golden vectors"
    which = sys.argv[1:] or ["snv", "cnv", "sv", "auto"]
    if "auto" in which:
        gen_auto()
    if "cnv" in which:
        gen_cnv()
    if "sv" in which:
        gen_sv()
    if "snv" in which:
        gen_snv()
