"""Generates the committed golden vectors by running the REFERENCE itself
(/root/reference, imported through tests/refshim stand-ins for cyvcf2/pysam).

Run in the authoring container only:   python tests/golden/make_golden.py
The GPU box never has /root/reference; tests read the JSON files written here.
Every fixture stores its inputs (inline, or as a synth.small configuration plus a
digest of the generated records) and the reference's outputs.
"""
import contextlib
import copy
import hashlib
import io
import itertools
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import refrun  # noqa: E402
from synth.small import SmallConfig, make_small  # noqa: E402
from unfazed_amd.model import SiteRecord  # noqa: E402

SNV_CASES = [
    ("default", dict(seed=11, n_dnms=10), dict()),
    ("no_extended", dict(seed=12, n_dnms=24), dict(no_extended=True)),
    ("find_many", dict(seed=13, n_dnms=10), dict(multithread_proc_min=1)),
    ("chr_prefix", dict(seed=14, n_dnms=8, chr_prefix="chr"), dict()),
    ("noisy_clustered", dict(seed=15, n_dnms=10, base_err=0.02, cluster_prob=1.0, lowq_prob=0.06), dict()),
    ("params", dict(seed=16, n_dnms=10), dict(min_gt_qual=30, min_depth=0, ab_het=[0.3, 0.7], search_dist=2000)),
    ("two_kids_odd_reads", dict(seed=17, n_dnms=12, kids=["kidA", "kidB"], odd_read_prob=0.15, softclip_prob=0.1,
                                indel_prob=0.08), dict()),
    ("indel_dnms", dict(seed=18, n_dnms=12, indel_dnm_frac=0.6, mnp_dnm_frac=0.2), dict()),
    ("read_goal", dict(seed=19, n_dnms=8, cluster_prob=1.0), dict(insert_size_max_sample=20)),
    ("readlen100", dict(seed=20, n_dnms=8), dict(readlen=100)),
    ("dense_sites", dict(seed=21, n_dnms=8, site_rate=1 / 150.0, cluster_prob=1.0, base_err=0.01), dict()),
    ("find_many_two_kids_noext", dict(seed=22, n_dnms=12, kids=["kidA", "kidB"]),
     dict(multithread_proc_min=1, no_extended=True)),
]


def dataset_digest(ds):
    h = hashlib.sha256()
    h.update(json.dumps(ds.samples).encode())
    for r in ds.sites:
        h.update(repr((r.chrom, r.start, r.ref, r.alts, list(map(int, r.gt_types)), list(map(int, r.ref_depths)),
                       list(map(int, r.alt_depths)), [float(x) for x in r.gt_quals])).encode())
    for kid in sorted(ds.reads):
        for s in ds.reads[kid]:
            h.update(repr((s.qname, s.flag, s.tid, s.pos, s.mapq, s.cigar, s.mtid, s.mpos, s.tlen, s.seq,
                           None if s.qual is None else bytes(s.qual), s.has_sa)).encode())
    h.update(json.dumps(ds.dnms, sort_keys=True).encode())
    return h.hexdigest()


def norm_records(recs):
    return {k: {kk: (sorted(v) if isinstance(v, list) else v) for kk, v in r.items()} for k, r in recs.items()}


def gen_snv():
    for name, cfgkw, runkw in SNV_CASES:
        ds = make_small(SmallConfig(**cfgkw))
        recs, dnms, err, cutoffs = refrun.run_phase_snvs(ds, tag=name, **runkw)
        for d in dnms:
            d["bam"] = "mem://%s.bam" % d["kid"]
        out = dict(
            config=cfgkw, run=runkw, digest=dataset_digest(ds),
            record_order=list(recs.keys()), records=norm_records(recs), dnms=dnms,
            stderr=err.splitlines(), cutoffs=cutoffs,
        )
        with open(os.path.join(HERE, "snv_%s.json" % name), "w") as fh:
            json.dump(out, fh, indent=0, sort_keys=True)
        print("snv", name, len(recs), "records")


def gen_grid():
    """G2/G3: find() over an exhaustive-ish genotype/depth/GQ grid, SNV mode and CNV (DEL, DUP) mode."""
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    rng = np.random.RandomState(5)
    samples = ["kid", "dad", "mom"]
    ped = {"kid": {"kid": "kid", "dad": "dad", "mom": "mom", "sex": "2"}}
    recs = []
    pos = 1000
    depth_pairs = [(0, 0), (-1, -1), (3, 2), (2, 3), (30, 0), (0, 30), (15, 15), (25, 5), (5, 25), (20, 10), (10, 20),
                   (28, 2), (2, 28), (9, 1), (6, 4), (3, 3), (40, 40), (2, 1), (8, 2), (33, 67), (67, 33), (1, 0)]
    gqs = [-1.0, 0.0, 19.0, 19.5, 20.0, 99.0]
    gts = [0, 1, 2, 3]
    rows = []
    for kg, dg, mg in itertools.product(gts, gts, gts):
        for _ in range(14):
            plausible = {0: [(30, 0), (28, 2), (9, 1), (40, 1)], 1: [(15, 15), (20, 10), (10, 20), (6, 4), (33, 67), (67, 33), (25, 5), (5, 25), (14, 30), (30, 14)],
                         2: [(15, 15)], 3: [(0, 30), (2, 28), (1, 40)]}
            d = [plausible[gg][rng.randint(len(plausible[gg]))] if rng.rand() < 0.7 else depth_pairs[rng.randint(len(depth_pairs))]
                 for gg in (kg, dg, mg)]
            q = [gqs[rng.randint(len(gqs))] if rng.rand() < 0.25 else 99.0 for _ in range(3)]
            rows.append(((kg, dg, mg), d, q))
    for (g, d, q) in rows:
        recs.append(SiteRecord("1", pos, "A", ["C"], list(g), [x[0] for x in d], [x[1] for x in d], list(q)))
        pos += 3
    # complex records
    recs.append(SiteRecord("1", pos, "A", ["C", "G"], [1, 0, 1], [15, 30, 15], [15, 0, 15], [99.0] * 3)); pos += 3
    recs.append(SiteRecord("1", pos, "AT", ["A"], [1, 0, 1], [15, 30, 15], [15, 0, 15], [99.0] * 3)); pos += 3
    recs.append(SiteRecord("1", pos, "A", ["*"], [1, 0, 1], [15, 30, 15], [15, 0, 15], [99.0] * 3)); pos += 3
    recs.append(SiteRecord("1", pos, "A", ["AT"], [1, 0, 1], [15, 30, 15], [15, 0, 15], [99.0] * 3)); pos += 3
    end = pos + 100
    cyvcf2.register("mem://grid.vcf", samples, recs)
    param_sets = [
        dict(ab_homref=[0.0, 0.2], ab_homalt=[0.8, 1.0], ab_het=[0.2, 0.8], min_gt_qual=20, min_depth=10),
        dict(ab_homref=[0.0, 0.1], ab_homalt=[0.9, 1.0], ab_het=[0.3, 0.7], min_gt_qual=0, min_depth=0),
        dict(ab_homref=[0.0, 0.34], ab_homalt=[0.66, 1.0], ab_het=[0.25, 0.75], min_gt_qual=19, min_depth=5),
    ]
    cases = []
    for ps in param_sets:
        for whole, vt, st, en, sd in ((False, "POINT", 2000, 2001, 100000), (True, "DEL", 900, end, 0),
                                     (True, "DUP", 900, end, 0), (True, "INV", 900, end, 0),
                                     (False, "POINT", 1300, 1310, 500)):
            dn = [{"chrom": "1", "start": st, "end": en, "kid": "kid", "vartype": vt, "bam": "", "cram_ref": None}]
            with warnings.catch_warnings(), contextlib.redirect_stderr(io.StringIO()):
                warnings.simplefilter("ignore")
                out = isf.find(dn, ped, "mem://grid.vcf", sd, 1, "38", 1000, True, ps["ab_homref"], ps["ab_homalt"],
                               ps["ab_het"], ps["min_gt_qual"], ps["min_depth"], whole_region=whole)
            cases.append(dict(params=ps, whole_region=whole, search_dist=sd, dnm=dict(start=st, end=en, vartype=vt),
                              candidate_sites=out[0]["candidate_sites"], het_sites=out[0]["het_sites"]))
    sites = [dict(start=r.start, ref=r.ref, alts=r.alts, gt=list(map(int, r.gt_types)), rd=list(map(int, r.ref_depths)),
                  ad=list(map(int, r.alt_depths)), gq=[float(x) for x in r.gt_quals]) for r in recs]
    with open(os.path.join(HERE, "find_grid.json"), "w") as fh:
        json.dump(dict(samples=samples, sites=sites, cases=cases), fh)
    print("grid", len(sites), "sites", len(cases), "cases")


def gen_bsearch():
    """G1: binary_search order and match_informative_sites drop decisions."""
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    rng = np.random.RandomState(9)
    cases = []
    fixed = [(100, 100, [100, 150]), (150, 250, [100, 150, 200, 250, 300]), (0, 10, []), (5, 5, [5]),
             (10, 20, [20]), (10, 20, [10, 20]), (10, 20, [5, 20, 20, 25]), (10, 20, [12, 12, 12])]
    for start, end, pos in fixed:
        m = ss.binary_search(start, end, [{"pos": p, "i": i} for i, p in enumerate(pos)])
        cases.append(dict(start=start, end=end, pos=pos, order=[x["i"] for x in m]))
    for _ in range(400):
        n = rng.randint(0, 41)
        pos = sorted(int(x) for x in rng.randint(0, 300, size=n))
        start = int(rng.randint(0, 300))
        end = start + int(rng.choice([0, 1, 5, 40, 151]))
        if n and rng.rand() < 0.4:
            end = pos[rng.randint(n)]
            start = max(0, end - int(rng.randint(0, 60)))
        m = ss.binary_search(start, end, [{"pos": p, "i": i} for i, p in enumerate(pos)])
        cases.append(dict(start=start, end=end, pos=pos, order=[x["i"] for x in m]))
    with open(os.path.join(HERE, "bsearch.json"), "w") as fh:
        json.dump(cases, fh)
    print("bsearch", len(cases))


def gen_summarize():
    """G7: summarize_record and the BED text over a grid of evidence counts."""
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    cases = []
    rng = np.random.RandomState(3)

    def rec(dr, mr, ds, ms, cd, cm, etype="readbacked", chrom="1", start=100):
        return {
            "region": {"chrom": chrom, "start": start, "end": start + 1}, "vartype": "POINT" if cd is None else "DEL",
            "kid": "kid", "dad": "dad", "mom": "mom",
            "dad_sites": [str(1000 + i * 7) for i in range(ds)], "mom_sites": [str(990 + i * 11) for i in range(ms)],
            "evidence_type": etype,
            "dad_reads": ["rd%d" % i for i in range(dr)], "mom_reads": ["rm%d" % i for i in range(mr)],
            "cnv_dad_sites": "" if cd is None else [str(2000 + i) for i in range(cd)],
            "cnv_mom_sites": "" if cm is None else [str(3000 + i) for i in range(cm)],
            "cnv_evidence_type": "" if cd is None else "ALLELE-BALANCE",
        }

    combos = []
    for dr, mr in itertools.product([0, 1, 2, 9, 10, 11, 12], repeat=2):
        combos.append((dr, mr, min(dr, 3), min(mr, 2), None, None))
    for dr, mr, cd, cm in itertools.product([0, 1, 10], [0, 1, 10], [0, 1, 5, 12], [0, 1, 5, 12]):
        combos.append((dr, mr, min(dr, 2), min(mr, 2), cd, cm))
    for (dr, mr, ds, ms, cd, cm) in combos:
        for ratio in (1, 2, 10):
            for amb in (False, True):
                r = rec(dr, mr, ds, ms, cd, cm, etype="readbacked" if cd is None else ("readbacked,ALLELE-BALANCE" if dr + mr else ""))
                out = uz.summarize_record(copy.deepcopy(r), amb, True, ratio)
                cases.append(dict(record=r, ratio=ratio, include_ambiguous=amb, summary=out))
    auto = {"region": {"chrom": "chrY", "start": 5, "end": 6}, "vartype": "POINT", "kid": "kid", "dad": "dad", "mom": "mom",
            "cnv_dad_sites": "NA", "cnv_mom_sites": "NA", "cnv_evidence_type": "SEX-CHROM", "dad_sites": "", "mom_sites": "",
            "evidence_type": "SEX-CHROM", "dad_reads": [], "mom_reads": []}
    autox = copy.deepcopy(auto)
    autox["region"]["chrom"] = "X"
    for r in (auto, autox):
        cases.append(dict(record=r, ratio=10, include_ambiguous=False, summary=uz.summarize_record(copy.deepcopy(r), False, True, 10)))
    # BED text
    beds = []
    for verbose, amb in itertools.product((False, True), (False, True)):
        records = {}
        for i, (dr, mr, ds, ms, cd, cm) in enumerate(combos[::7]):
            r = rec(dr, mr, ds, ms, cd, cm, chrom=str(rng.choice(["1", "2", "10", "X"])), start=int(rng.randint(1, 1000)))
            records["k%d" % i] = r
        records["auto"] = auto
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            uz.write_bed_output(copy.deepcopy(records), amb, verbose, "/dev/stdout", 10)
        beds.append(dict(records=records, verbose=verbose, include_ambiguous=amb, lines=buf.getvalue().splitlines()))
    with open(os.path.join(HERE, "summarize.json"), "w") as fh:
        json.dump(dict(cases=cases, beds=beds), fh)
    print("summarize", len(cases), "cases", len(beds), "beds")


def gen_cutoff():
    """G9: estimate_concordant_insert_len."""
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    from unfazed_amd.model import Segment
    rng = np.random.RandomState(4)
    cases = []
    for n, readlen, cap in ((1, 151, 100), (2, 151, 100), (37, 151, 100), (500, 151, 100), (500, 100, 1000),
                            (2000, 151, 1000000), (199, 151, 1000000), (200, 151, 1000000), (201, 151, 1000000)):
        tl = [int(x) for x in np.clip(rng.normal(450, 80, size=n), -900, 2000).astype(int) * rng.choice([-1, 1], size=n)]
        if n > 50:
            tl[7] = 250000
            tl[9] = 0
        segs = [Segment("q%d" % i, 0, 0, i, 60, [(0, 10)], 0, 0, t, "A" * 10, [30] * 10) for i, t in enumerate(tl)]
        pysam.register("mem://cut.bam", ["1"], segs)
        rc.READLEN = readlen
        v = rc.estimate_concordant_insert_len(pysam.AlignmentFile("mem://cut.bam"), cap, 3)
        cases.append(dict(tlen=tl, readlen=readlen, insert_size_max_sample=cap, cutoff=float(v)))
    with open(os.path.join(HERE, "cutoff.json"), "w") as fh:
        json.dump(cases, fh)
    print("cutoff", len(cases))


def make_cnv_dataset(seed=31, n=10):
    """synth.small trio plus DEL / DUP / INV events laid over the site-rich windows, with the kid's
    genotypes / depths inside the events re-drawn as a hemizygous deletion or a 2:1 duplication."""
    rng = np.random.RandomState(seed)
    ds = make_small(SmallConfig(seed=seed, n_dnms=n, coverage_per_hap=1.0, site_rate=1 / 250.0))
    col = {s: i for i, s in enumerate(ds.samples)}
    k = col["kid1"]
    svs = []
    for i, d in enumerate(ds.dnms):
        vt = ["DEL", "DUP", "DEL", "DUP", "INV"][i % 5]
        st, en = d["start"] - int(rng.randint(500, 4000)), d["start"] + int(rng.randint(500, 4000))
        svs.append({"chrom": d["chrom"], "start": st, "end": en, "kid": "kid1", "vartype": vt, "bam": "", "cram_ref": None})
        for r in ds.sites:
            if r.chrom != d["chrom"] or not (st <= r.start <= en):
                continue
            u = rng.rand()
            if vt == "DEL" and u < 0.7:
                depth = int(rng.randint(3, 25))
                if rng.rand() < 0.5:
                    r.gt_types[k], r.ref_depths[k], r.alt_depths[k] = 0, depth, 0
                else:
                    r.gt_types[k], r.ref_depths[k], r.alt_depths[k] = 3, 0, depth
            elif vt == "DUP" and u < 0.7:
                a, b = [(20, 10), (10, 20), (30, 14), (14, 30), (22, 11), (3, 9), (16, 15), (67, 33)][rng.randint(8)]
                r.gt_types[k], r.ref_depths[k], r.alt_depths[k] = 1, a, b
    # two events of the same kid sharing a start (find_many multiplicity), one tiny event
    svs.append(dict(svs[0]))
    svs.append({"chrom": ds.dnms[1]["chrom"], "start": ds.dnms[1]["start"] - 5, "end": ds.dnms[1]["start"] + 7, "kid": "kid1",
                "vartype": "DEL", "bam": "", "cram_ref": None})
    ds.dnms = svs
    return ds


def gen_cnv():
    """G8: run_cnv_phasing (allele-balance phasing of DEL / DUP) through find()."""
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    ds = make_cnv_dataset()
    vcf, bams = refrun.register(ds, "cnv")
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
                          dnms=dnms, summaries=summaries))
        print("cnv", len(recs), "records")
    with open(os.path.join(HERE, "cnv.json"), "w") as fh:
        json.dump(dict(digest=dataset_digest(ds), cases=cases), fh)


SV_CASES = [
    ("default", dict(seed=41, n_svs=8), dict()),
    ("no_extended", dict(seed=42, n_svs=8), dict(no_extended=True)),
    ("params", dict(seed=43, n_svs=8), dict(search_dist=2000, min_gt_qual=10, split_error_margin=0)),
]


def gen_sv():
    """phase_svs end to end: allele-balance + read-backed SV evidence (collect_reads_sv) and their merge."""
    from synth.small_sv import SvConfig, make_small_sv
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    for name, cfgkw, runkw in SV_CASES:
        ds = make_small_sv(SvConfig(**cfgkw))
        recs, dnms, err, cutoffs = refrun.run_phase_svs(ds, tag="sv_" + name, **runkw)
        for d in dnms:
            d["bam"] = "mem://%s.bam" % d["kid"]
        summaries = {k: uz.summarize_record(copy.deepcopy(r), True, False, 10) for k, r in recs.items()}
        out = dict(config=cfgkw, run=runkw, digest=dataset_digest(ds), record_order=list(recs.keys()),
                   records=norm_records(recs), dnms=dnms, stderr=err.splitlines(), summaries=summaries)
        with open(os.path.join(HERE, "sv_%s.json" % name), "w") as fh:
            json.dump(out, fh, sort_keys=True)
        print("sv", name, len(recs), "records", sum(1 for r in recs.values() if r["dad_reads"] or r["mom_reads"]), "read-backed")


CLI_CFG = dict(seed=77, n_dnms=10, kids=["kidA", "kidB"], odd_read_prob=0.1)


def gen_cli():
    """Whole driver (reference unfazed.unfazed(args)): BED text and the per-sample GT / UOPS / UET of the
    annotated VCF, for VCF and BED DNM input."""
    import argparse
    import tempfile
    from filesio import dump_dataset
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    ds = make_small(SmallConfig(**CLI_CFG))
    tmp = tempfile.mkdtemp()
    paths = dump_dataset(ds, tmp)
    cyvcf2.register(paths["sites"], ds.samples, ds.sites)
    want = {(d["chrom"], d["start"]) for d in ds.dnms}
    seen, dn_recs = set(), []
    for r in ds.sites:
        if (r.chrom, r.start) in want and (r.chrom, r.start) not in seen:
            seen.add((r.chrom, r.start))
            r.genotypes = [[0, 1, False] if g == 1 else ([1, 1, False] if g == 3 else ([-1, -1, False] if g == 2 else [0, 0, False]))
                           for g in r.gt_types]
            dn_recs.append(r)
    cyvcf2.register(paths["dnm_vcf"], ds.samples, dn_recs)
    for kid, segs in ds.reads.items():
        pysam.register(paths["bams"][kid], ds.contigs, segs)
    runs = []
    for dnm_key, out_type, amb, verbose in (("dnm_vcf", "bed", False, False), ("dnm_vcf", "bed", True, True),
                                            ("dnm_bed", "bed", True, False), ("dnm_vcf", "vcf", False, False),
                                            ("dnm_vcf", "vcf", True, False)):
        args = argparse.Namespace(
            dnms=paths[dnm_key], sites=paths["sites"], ped=paths["ped"], bam_dir=None,
            bam_pairs=[[k, v] for k, v in paths["bams"].items()], threads=1, output_type=out_type,
            include_ambiguous=amb, verbose=verbose, outfile="/dev/stdout", reference=None, build="38",
            no_extended=False, multiread_proc_min=1000, quiet=True, min_gt_qual=20, min_depth=10,
            ab_homref=[0.0, 0.2], ab_homalt=[0.8, 1.0], ab_het=[0.2, 0.8], evidence_min_ratio=10, search_dist=5000,
            insert_size_max_sample=1000000, min_map_qual=1, stdevs=3, readlen=151, split_error_margin=5, max_reads=100)
        sp.concordant_upper_lens.clear()
        buf = io.StringIO()
        writers = []
        orig_writer = uz.Writer

        class W(orig_writer):
            def __init__(self, *a, **k):
                orig_writer.__init__(self, *a, **k)
                writers.append(self)
        uz.Writer = W
        try:
            with warnings.catch_warnings(), contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
                warnings.simplefilter("ignore")
                uz.unfazed(args)
        finally:
            uz.Writer = orig_writer
        run = dict(dnms=dnm_key, output_type=out_type, include_ambiguous=amb, verbose=verbose)
        if out_type == "bed":
            run["lines"] = buf.getvalue().splitlines()
        else:
            body = []
            for v in writers[0].records:
                body.append(dict(chrom=v.CHROM, pos=v.POS, genotypes=[[int(g[0]), int(g[1]), bool(g[2])] for g in v.genotypes],
                                 uops=[float(x) for x in v.formats["UOPS"]], uet=[float(x) for x in v.formats["UET"]]))
            run["body"] = body
            run["header_added"] = writers[0].tmpl.header_lines + [d["ID"] for d in writers[0].tmpl.formats_added]
        runs.append(run)
        print("cli", dnm_key, out_type, amb, len(run.get("lines", run.get("body"))))
    with open(os.path.join(HERE, "cli.json"), "w") as fh:
        json.dump(dict(config=CLI_CFG, digest=dataset_digest(ds), runs=runs), fh)


if __name__ == "__main__":
    assert refrun.available(), "/root/reference is required to generate golden vectors"
    which = sys.argv[1:] or ["snv", "grid", "bsearch", "summarize", "cutoff", "cnv"]
    if "cnv" in which:
        gen_cnv()
    if "sv" in which or not sys.argv[1:]:
        gen_sv()
    if "cli" in which or not sys.argv[1:]:
        gen_cli()
    if "snv" in which:
        gen_snv()
    if "grid" in which:
        gen_grid()
    if "bsearch" in which:
        gen_bsearch()
    if "summarize" in which:
        gen_summarize()
    if "cutoff" in which:
        gen_cutoff()
