"""Backend that answers the PhasingHost through the CPU oracle -- tests only.
(The product backend is unfazed_amd.engine.HipEngine; nothing under unfazed_amd
imports this file.)"""
import numpy as np

from oracle import oracle as orc
from unfazed_amd import abi


class OracleBackend:
    def upload_sites(self, sites):
        return abi.sites_view(sites)

    def add_family(self, sites_h, gt, rd, ad, gq, wide=None):
        return (sites_h, abi.family_view(gt, rd, ad, gq, wide))

    def upload_reads(self, reads, min_base_qual=None, point_only=False, fetches=None, all_bases=False, wide_no_units=False):
        return abi.reads_view(reads)

    def classify(self, fam, params):
        sites_h, fv = fam
        return orc.classify(params, sites_h, fv)

    def find(self, fam, dv, params, mode):
        sites_h, fv = fam
        return orc.find(params, sites_h, fv, dv, mode)

    def phase(self, fam, reads_h, dv, params, found_list, want_lists=True, find_mode=2):
        sites_h, fv = fam
        n = dv.view.n
        co = np.zeros(n + 1, dtype=np.int64)
        ho = np.zeros(n + 1, dtype=np.int64)
        for k, f in enumerate(found_list):
            co[k + 1] = co[k] + len(f["cand_idx"])
            ho[k + 1] = ho[k] + len(f["het_idx"])
        ci = np.concatenate([f["cand_idx"] for f in found_list] + [np.zeros(0, np.int32)]).astype(np.int32)
        cf = np.concatenate([f["cand_flags"] for f in found_list] + [np.zeros(0, np.uint8)]).astype(np.uint8)
        hi = np.concatenate([f["het_idx"] for f in found_list] + [np.zeros(0, np.int32)]).astype(np.int32)
        r = orc.phase(params, sites_h, reads_h, dv, (co, ci, cf, ho, hi), keep_lists=True)
        lists = []
        vo, vv = r["vote_off"], r["vote_val"]
        for k in range(n):
            lists.append(tuple(vv[vo[4 * k + j]: vo[4 * k + j + 1]] for j in range(4)))
        r["lists"] = lists if want_lists else None
        return r

    def phase_cnv(self, fam, dv, params, rb_counts=None, want_lists=True):
        sites_h, fv = fam
        return orc.phase_cnv(params, sites_h, fv, dv, rb_counts)

    def phase_cohort(self, groups, dv, params, found_list, want_lists=True, find_mode=2):
        """the cohort batch kid by kid (the oracle has no notion of a batch)"""
        n = dv.view.n
        out = dict(status=np.zeros(n, np.int32), counts=np.zeros((n, 4), np.int32), origin=np.zeros(n, np.int32),
                   evidence=np.zeros(n, np.int32), lists=[None] * n if want_lists else None)
        a = dv.arrays
        for fam, rh, first, count, cutoff in groups:
            sl = slice(first, first + count)
            off = a["allele_off"]
            refs = [bytes(a["alleles"][off[2 * k]: off[2 * k + 1]]) for k in range(first, first + count)]
            alts = [bytes(a["alleles"][off[2 * k + 1]: off[2 * k + 2]]) for k in range(first, first + count)]
            sub = abi.dnms_view(a["contig"][sl], a["rcontig"][sl], a["start"][sl], a["end"][sl], a["vartype"][sl], refs, alts, cutoff,
                                dflags=a["dflags"][sl], mult=a["mult"][sl])
            r = self.phase(fam, rh, sub, params, found_list[first: first + count], want_lists, find_mode)
            for k in ("status", "counts", "origin", "evidence"):
                out[k][sl] = r[k]
            if want_lists:
                out["lists"][sl] = r["lists"]
        return out
