"""Golden vectors for sites whose allele depths do not fit 16 bits (the reference takes any depth:
informative_site_finder.py:46-73, :76-134): the reference's find() over a genotype x depth grid that mixes ordinary depths with
32768, 70000, 10^6 ..., in SNV mode and in whole-region (DEL / DUP) mode, for three threshold sets.
Run in the authoring container only:   python tests/golden/make_golden_deep.py   -> tests/golden/find_grid_deep.json"""
import contextlib
import io
import itertools
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refrun  # noqa: E402
from unfazed_amd.model import SiteRecord  # noqa: E402


def main():
    cyvcf2, pysam, isf, rc, ss, sp, svp, uz = refrun._import()
    rng = np.random.RandomState(77)
    samples = ["kid", "dad", "mom"]
    ped = {"kid": {"kid": "kid", "dad": "dad", "mom": "mom", "sex": "2"}}
    deep = {0: [(32768, 0), (70000, 30), (1000000, 9000), (40000, 8000), (32767, 1), (65535, 0), (65536, 13000), (200000, 60000)],
            1: [(32768, 32768), (70000, 70000), (1000000, 1000000), (40000, 9000), (9000, 40000), (32768, 10000), (16384, 16384), (70000, 23000),
                (23000, 70000), (1000000, 490000), (33000, 67000), (67000, 33000)],
            2: [(32768, 32768), (70000, 0)],
            3: [(0, 32768), (30, 70000), (9000, 1000000), (1, 32767), (8000, 40000), (60000, 200000)]}
    plain = {0: [(30, 0), (28, 2), (9, 1)], 1: [(15, 15), (20, 10), (10, 20), (6, 4), (25, 5)], 2: [(15, 15)], 3: [(0, 30), (2, 28)]}
    odd = [(0, 0), (-1, -1), (3, 2), (32768, -1), (-1, 70000), (0, 32768), (32768, 0), (4, 70000), (70000, 2)]
    gqs = [-1.0, 19.0, 20.0, 99.0]
    recs, pos = [], 1000
    for kg, dg, mg in itertools.product([0, 1, 2, 3], repeat=3):
        for _ in range(10):
            d = []
            for gg in (kg, dg, mg):
                u = rng.rand()
                pool = deep[gg] if u < 0.55 else (plain[gg] if u < 0.85 else odd)
                d.append(pool[rng.randint(len(pool))])
            if not any(max(x) > 32767 for x in d):  # every record of this grid is deep in at least one member
                k = rng.randint(3)
                gg = (kg, dg, mg)[k]
                d[k] = deep[gg][rng.randint(len(deep[gg]))]
            q = [gqs[rng.randint(len(gqs))] if rng.rand() < 0.2 else 99.0 for _ in range(3)]
            recs.append(SiteRecord("1", pos, "A", ["C"], [kg, dg, mg], [x[0] for x in d], [x[1] for x in d], q))
            pos += 3
    recs.append(SiteRecord("1", pos, "A", ["C", "G"], [1, 0, 1], [70000, 30, 15], [70000, 0, 15], [99.0] * 3)); pos += 3  # complex + deep
    end = pos + 100
    cyvcf2.register("mem://deep.vcf", samples, recs)
    param_sets = [
        dict(ab_homref=[0.0, 0.2], ab_homalt=[0.8, 1.0], ab_het=[0.2, 0.8], min_gt_qual=20, min_depth=10),
        dict(ab_homref=[0.0, 0.1], ab_homalt=[0.9, 1.0], ab_het=[0.3, 0.7], min_gt_qual=0, min_depth=0),
        dict(ab_homref=[0.0, 0.34], ab_homalt=[0.66, 1.0], ab_het=[0.25, 0.75], min_gt_qual=19, min_depth=40000),
    ]
    cases = []
    for ps in param_sets:
        for whole, vt, st, en, sd in ((False, "POINT", 2000, 2001, 100000), (True, "DEL", 900, end, 0), (True, "DUP", 900, end, 0),
                                     (False, "POINT", 1300, 1310, 500)):
            dn = [{"chrom": "1", "start": st, "end": en, "kid": "kid", "vartype": vt, "bam": "", "cram_ref": None}]
            with warnings.catch_warnings(), contextlib.redirect_stderr(io.StringIO()):
                warnings.simplefilter("ignore")
                out = isf.find(dn, ped, "mem://deep.vcf", sd, 1, "38", 1000, True, ps["ab_homref"], ps["ab_homalt"], ps["ab_het"], ps["min_gt_qual"],
                               ps["min_depth"], whole_region=whole)
            cases.append(dict(params=ps, whole_region=whole, search_dist=sd, dnm=dict(start=st, end=en, vartype=vt),
                              candidate_sites=out[0]["candidate_sites"], het_sites=out[0]["het_sites"]))
    sites = [dict(start=r.start, ref=r.ref, alts=r.alts, gt=list(map(int, r.gt_types)), rd=list(map(int, r.ref_depths)), ad=list(map(int, r.alt_depths)),
                  gq=[float(x) for x in r.gt_quals]) for r in recs]
    with open(os.path.join(HERE, "find_grid_deep.json"), "w") as fh:
        json.dump(dict(samples=samples, sites=sites, cases=cases), fh)
    print("deep grid", len(sites), "sites", len(cases), "cases", sum(len(c["candidate_sites"]) for c in cases), "candidates", sum(len(c["het_sites"]) for c in cases), "het sites")


if __name__ == "__main__":
    main()
