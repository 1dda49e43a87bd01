"""Runs the REFERENCE (/root/reference, imported through tests/refshim stand-ins
for cyvcf2/pysam) on a synth.small dataset.  Only usable in the authoring
container; used by tests/golden/make_golden.py and make_golden_wide.py to produce the
committed vectors and by tests/golden/crosscheck.py (a script, not a test: the reference
against the oracle on fresh seeds)."""
import contextlib
import copy
import io
import os
import sys
import warnings

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, "unfazed"))


def _import():
    shim = os.path.join(_HERE, "refshim")
    if shim not in sys.path:
        sys.path.insert(0, shim)
    if REF not in sys.path:
        sys.path.append(REF)
    import cyvcf2  # noqa: F401  (the stand-in)
    import pysam  # noqa: F401
    import unfazed.informative_site_finder as isf
    import unfazed.read_collector as rc
    import unfazed.site_searcher as ss
    import unfazed.snv_phaser as sp
    import unfazed.sv_phaser as svp
    import unfazed.unfazed as uz
    return cyvcf2, pysam, isf, rc, ss, sp, svp, uz


DEFAULTS = dict(
    threads=1, build="38", no_extended=False, multithread_proc_min=1000, quiet_mode=False,
    ab_homref=[0.0, 0.2], ab_homalt=[0.8, 1.0], ab_het=[0.2, 0.8], min_gt_qual=20, min_depth=10,
    search_dist=5000, insert_size_max_sample=1000000, stdevs=3, min_map_qual=1, readlen=151,
    split_error_margin=5,
)


def register(ds, tag="ds"):
    cyvcf2, pysam, *_ = _import()
    vcf = "mem://%s/sites.vcf" % tag
    cyvcf2.register(vcf, ds.samples, ds.sites)
    bams = {}
    for kid, segs in ds.reads.items():
        bams[kid] = "mem://%s/%s.bam" % (tag, kid)
        pysam.register(bams[kid], ds.contigs, segs)
    return vcf, bams


def run_phase_snvs(ds, tag="ds", **kw):
    """-> (records, annotated dnm list, stderr text)"""
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = _import()
    a = dict(DEFAULTS)
    a.update(kw)
    vcf, bams = register(ds, tag)
    dnms = copy.deepcopy(ds.dnms)
    for d in dnms:
        d["bam"] = bams[d["kid"]]
    sp.concordant_upper_lens.clear()
    err = io.StringIO()
    with warnings.catch_warnings(), contextlib.redirect_stderr(err):
        warnings.simplefilter("ignore")
        recs = sp.phase_snvs(
            dnms, list(ds.pedigrees), ds.pedigrees, vcf, a["threads"], a["build"], a["no_extended"],
            a["multithread_proc_min"], a["quiet_mode"], a["ab_homref"], a["ab_homalt"], a["ab_het"],
            a["min_gt_qual"], a["min_depth"], a["search_dist"], a["insert_size_max_sample"], a["stdevs"],
            a["min_map_qual"], a["readlen"], a["split_error_margin"],
        )
    return recs, dnms, err.getvalue(), dict(sp.concordant_upper_lens)


def run_find(ds, tag="ds", whole_region=False, **kw):
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = _import()
    a = dict(DEFAULTS)
    a.update(kw)
    vcf, bams = register(ds, tag)
    dnms = copy.deepcopy(ds.dnms)
    err = io.StringIO()
    with warnings.catch_warnings(), contextlib.redirect_stderr(err):
        warnings.simplefilter("ignore")
        out = isf.find(
            dnms, ds.pedigrees, vcf, a["search_dist"], a["threads"], a["build"], a["multithread_proc_min"],
            a["quiet_mode"], a["ab_homref"], a["ab_homalt"], a["ab_het"], a["min_gt_qual"], a["min_depth"],
            whole_region=whole_region,
        )
    return out, err.getvalue()


def run_phase_svs(ds, tag="sv", **kw):
    """-> (records, annotated dnm list, stderr text)"""
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = _import()
    a = dict(DEFAULTS)
    a.update(kw)
    vcf, bams = register(ds, tag)
    dnms = copy.deepcopy(ds.dnms)
    for d in dnms:
        d["bam"] = bams[d["kid"]]
    svp.concordant_upper_lens.clear()
    err = io.StringIO()
    with warnings.catch_warnings(), contextlib.redirect_stderr(err):
        warnings.simplefilter("ignore")
        recs = svp.phase_svs(
            dnms, list(ds.pedigrees), ds.pedigrees, vcf, a["threads"], a["build"], a["no_extended"],
            a["multithread_proc_min"], a["quiet_mode"], a["ab_homref"], a["ab_homalt"], a["ab_het"],
            a["min_gt_qual"], a["min_depth"], a["search_dist"], a["insert_size_max_sample"], a["stdevs"],
            a["min_map_qual"], a["readlen"], a["split_error_margin"],
        )
    return recs, dnms, err.getvalue(), dict(svp.concordant_upper_lens)
