"""Host side of the per-DNM phasing path: everything the reference does around
the hot loops that is NOT arithmetic over sites/reads -- DNM bookkeeping,
contig-name resolution, the REF/ALT lookup, autophasing, messages, list order
and the `records` schema.  The arithmetic itself is done
by a *backend* object over the decoded column tables; the product backend is
unfazed_amd.engine.HipEngine (HIP kernels behind the C ABI).

Mirrors, by name and argument meaning:
  find / find_many      reference informative_site_finder.py:167-344 / :601-661
  run_read_phasing      reference snv_phaser.py:206-299
  run_cnv_phasing       reference sv_phaser.py:357-423
"""
from __future__ import annotations

import sys
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import abi
from .model import SFLAG_COMPLEX, ReadsTable, SitesTable

SEX_KEY = {"male": 1, "female": 2}  # reference utils.py:6
SV_TYPES = ["DEL", "DUP", "INV", "CNV", "DUP:TANDEM", "DEL:ME", "CPX", "CTX"]  # utils.py:8
SNV_TYPES = ["POINT", "SNV", "INDEL"]  # utils.py:9
# reference utils.py:26-43
grch37_par1 = {"x": [10001, 2781479], "y": [10001, 2781479]}
grch37_par2 = {"x": [155701383, 156030895], "y": [56887903, 57217415]}
grch38_par1 = {"x": [60001, 2699520], "y": [10001, 2649520]}
grch38_par2 = {"x": [154931044, 155260560], "y": [59034050, 59363566]}


_CHR = [chr(v) for v in range(256)]
_CHR_OBJ = np.array(_CHR, dtype=object)  # (a column of base bytes -> a list of one-character strings in one fancy index)


def get_prefix(sites: SitesTable) -> str:
    """reference utils.py:46-52 -- decided by the first record of the sites FILE: a table decoded through the tabix index for a
    batch's windows carries the file-level answer (`file_prefix`, from the index's first sequence name) instead of guessing it
    from the subset it happens to hold."""
    fp = getattr(sites, "file_prefix", None)
    if fp is not None:
        return fp
    if sites.n_sites == 0:
        return ""
    chrom = sites.contigs[0]
    return chrom[:3] if "chr" in chrom.lower() else ""


def autophaseable(denovo: dict, pedigrees: dict, build: str) -> bool:
    """reference informative_site_finder.py:137-164 (same test as snv_phaser.py:302-328)."""
    chrom = denovo["chrom"].lower().strip("chr")
    if chrom not in ["y", "x"]:
        return False
    if int(pedigrees[denovo["kid"]]["sex"]) != SEX_KEY["male"]:
        return False
    if build not in ["37", "38"]:
        return False
    par1, par2 = (grch37_par1, grch37_par2) if build == "37" else (grch38_par1, grch38_par2)
    if (
        par1[chrom][0] <= denovo["start"] <= par1[chrom][1]
        or par2[chrom][0] <= denovo["start"] <= par2[chrom][1]
    ):
        return False
    return True


def vartype_code(vartype: str) -> int:
    if vartype == "DEL":
        return abi.VT_DEL
    if vartype == "DUP":
        return abi.VT_DUP
    if vartype in SV_TYPES:
        return abi.VT_OTHER_SV
    return abi.VT_POINT


def concordant_cutoff(tlen_head: np.ndarray, readlen: int, stdevs: int) -> float:
    """reference read_collector.py:11-25, literally (quirk Q8: the percentile
    overwrites the array, so the stdev is 0 and --stdevs has no effect)."""
    insert_sizes = np.abs(np.asarray(tlen_head, dtype=np.int64) - (readlen * 2))
    insert_sizes = np.percentile(insert_sizes, 99.5)
    frag_len = int(np.mean(insert_sizes))
    stdev = np.std(insert_sizes)
    return frag_len + (stdev * stdevs)


class _Log:
    def __init__(self, quiet: bool):
        self.quiet = quiet

    def __call__(self, msg: str):
        if not self.quiet:
            print(msg, file=sys.stderr)


class _Sec:
    """development aid (UZ_HOST_TRACE=1): wall and per-thread CPU seconds of a named section of the host path, summed over calls"""
    acc: Dict[str, list] = {}
    on = bool(__import__("os").environ.get("UZ_HOST_TRACE"))

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _Sec.on:
            import time
            self.t = (time.perf_counter(), time.thread_time())
        return self

    def __exit__(self, *exc):
        if _Sec.on:
            import time
            a = _Sec.acc.setdefault(self.name, [0.0, 0.0, 0])
            a[0] += time.perf_counter() - self.t[0]
            a[1] += time.thread_time() - self.t[1]
            a[2] += 1
        return False

    @staticmethod
    def report():
        if _Sec.on and _Sec.acc:
            print("[uz] host sections (wall s, thread CPU s, calls): " + ", ".join("%s %.3f %.3f %d" % (k, v[0], v[1], v[2]) for k, v in _Sec.acc.items()), file=sys.stderr)
            _Sec.acc.clear()


class _VoteLists:
    """The four vote lists of every DNM of a chunk (uz_phase_votes: offsets [4n + 1], values) behind the indexing the host code uses --
    lists[k] -> (dad_reads, mom_reads, dad_sites, mom_sites) as arrays -- without 4n array slices made up front: the values become ONE Python
    list, cut per DNM when its record is built."""

    def __init__(self, vo, vv):
        self.vo, self.vv = vo, vv
        self._vol = vo.tolist()
        self._vvl = None

    def __len__(self):
        return (len(self._vol) - 1) // 4

    def __getitem__(self, k):
        o = self._vol
        return tuple(self.vv[o[4 * k + q]: o[4 * k + q + 1]] for q in range(4))

    def py(self, k):
        if self._vvl is None:
            self._vvl = self.vv.tolist()
        o, v = self._vol, self._vvl
        return v[o[4 * k]: o[4 * k + 1]], v[o[4 * k + 1]: o[4 * k + 2]], v[o[4 * k + 2]: o[4 * k + 3]], v[o[4 * k + 3]: o[4 * k + 4]]

    def strings(self, take):
        """Every value of the chunk's lists as the string a record carries -- a read's name (take(ids) -> names) or a position's decimal form --
        in ONE Python list (the lists of a DNM are slices of it): a fancy index over the distinct names and one str() per distinct position
        instead of a dictionary look-up and an append per list entry"""
        vo, vv = self.vo, self.vv
        n4 = vo.size - 1
        if vv.size == 0:
            return []
        is_read = np.repeat((np.arange(n4) & 3) < 2, np.diff(vo))
        out = np.empty(vv.size, object)
        ids = vv[is_read].astype(np.int64)
        if ids.size:
            u, inv = np.unique(ids, return_inverse=True)
            out[is_read] = np.array(take(u), object)[inv]
        pos = vv[~is_read]
        if pos.size:
            u, inv = np.unique(pos, return_inverse=True)
            out[~is_read] = np.array(list(map(str, u.tolist())), object)[inv]
        return out.tolist()

    def read_ids(self, ks):
        """the distinct name ids in the read lists of DNMs ks"""
        o = self.vo
        ks = np.asarray(ks, np.int64)
        a, b = o[4 * ks], o[4 * ks + 2]  # (the two read lists of a DNM lie end to end)
        n = (b - a).astype(np.int64)
        if int(n.sum()) == 0:
            return np.zeros(0, np.int64)
        idx = np.repeat(a - np.concatenate([[0], np.cumsum(n)[:-1]]), n) + np.arange(int(n.sum()))
        return np.unique(self.vv[idx].astype(np.int64))


class PhasingHost:
    """Drives one backend over one sites table and the reads tables of the kids."""

    def __init__(self, backend, sites: SitesTable, reads_by_bam: Dict[str, ReadsTable]):
        self.backend = backend
        self.sites = sites
        self.reads_by_bam = reads_by_bam
        self.prefix = get_prefix(sites)
        self._sites_h = backend.upload_sites(sites)
        self._fam_h: Dict[tuple, object] = {}
        self._reads_h: Dict[str, object] = {}
        self.cutoffs: Dict[str, float] = {}  # reference snv_phaser.py:14 concordant_upper_lens
        self.capacity_skipped: List[str] = []  # record keys the device refused (UZ_ST_CAPACITY)

    # ------------------------------------------------------------ handles
    def family(self, kid: str, dad: str, mom: str):
        key = (kid, dad, mom)
        if key not in self._fam_h:
            gt, rd, ad, gq = self.sites.family_columns(kid, dad, mom)
            wide = getattr(self.sites, "wide_depths", None)  # sites deeper than the 16-bit columns hold: their 32-bit depths
            self._fam_h[key] = self.backend.add_family(self._sites_h, gt, rd, ad, gq, wide=wide) if wide is not None else \
                self.backend.add_family(self._sites_h, gt, rd, ad, gq)
        return self._fam_h[key]

    def reads(self, bam: str, min_base_qual: int):
        """device handle of the whole table of a BAM (kept for the life of the host), staged for this base-quality threshold"""
        key = (bam, int(min_base_qual))
        if key not in self._reads_h:
            self._reads_h[key] = self.backend.upload_reads(self.reads_by_bam[bam], min_base_qual=int(min_base_qual))
        return self._reads_h[key]

    def _indexed(self, bam: str) -> bool:
        """does the reads source decode regions of this file through an index (session._LazyReads)?  (asked once per file and batch: the
        answer looks at the file system)"""
        memo = getattr(self, "_indexed_memo", None)
        if memo is not None and bam in memo:
            return memo[bam]
        f = getattr(self.reads_by_bam, "indexed", None)
        r = bool(f and f(bam))
        if memo is not None:
            memo[bam] = r
        return r

    def reads_header(self, bam: str) -> ReadsTable:
        """what the host logic needs before any record is decoded: contig names and the head of the file (insert cutoff)"""
        if self._indexed(bam):
            return self.reads_by_bam.header(bam)
        return self.reads_by_bam[bam]

    # --------------------------------------------------------------- find
    def find(
        self,
        dnms: List[dict],
        pedigrees: dict,
        search_dist: int,
        threads: int,
        build: str,
        multithread_proc_min: int,
        quiet_mode: bool,
        params: abi.Params,
        whole_region: bool = True,
        attach: bool = True,
        fetch: bool = True,
        defer_attach: bool = False,
    ):
        """informative_site_finder.find: annotates the DNM dicts in place with
        `candidate_sites` / `het_sites` and returns the list in the reference's order.
        Also returns, per returned DNM, the raw index lists (for the read stage)."""
        log = _Log(quiet_mode)
        many = len(dnms) >= multithread_proc_min  # :206
        if len(dnms) <= 0:
            return None, {}
        sites = self.sites
        sample_set = set(sites.samples)
        params.search_dist = int(search_dist)

        if many:
            order, auto_tail = self._find_many_order(dnms, pedigrees, build)
        else:
            order, auto_tail = list(range(len(dnms))), []

        # which DNMs get scanned, and with which sites-contig
        scan: List[int] = []
        contig_of: Dict[int, int] = {}
        mult_of: Dict[int, int] = {}
        dead_chroms = set()
        if many:
            mults = self._find_many_multiplicity(dnms, order)
            if whole_region:
                dead_chroms = self._find_many_keyerror_chroms(dnms, order, threads)
        for i in order:
            dn = dnms[i]
            if not many and autophaseable(dn, pedigrees, build):  # :216-217
                continue
            kid = dn["kid"]
            dad, mom = pedigrees[kid]["dad"], pedigrees[kid]["mom"]
            missing = False
            for s in (kid, dad, mom):
                if s not in sample_set:
                    # the reference prints the un-formatted template (:226, :430)
                    log("{} missing from SNV vcf/bcf" if not many else "{} missing from SNV bcf")
                    missing = True
            if missing:
                continue
            name = self.prefix + dn["chrom"].strip("chr")  # :18,:27 (quirk Q2)
            if many and name != dn["chrom"]:
                c = -1  # find_many keys its look-ups by the literal chrom (quirk Q3)
            else:
                c = sites.contig_index.get(name, -1)
            if many and dn["chrom"] in dead_chroms:
                c = -1
            scan.append(i)
            contig_of[i] = c
            mult_of[i] = mults[i] if many else 1

        found: Dict[int, dict] = {}
        pending_attach: list = []
        by_kid: Dict[str, List[int]] = {}
        for i in scan:
            by_kid.setdefault(dnms[i]["kid"], []).append(i)
        mode = (abi.FIND_WHOLE_REGION if whole_region else 0) | (0 if many else abi.FIND_SECOND_WINDOW)
        for kid, idxs in by_kid.items():
            dad, mom = pedigrees[kid]["dad"], pedigrees[kid]["mom"]
            fam = self.family(kid, dad, mom)
            n = len(idxs)
            dv = abi.dnms_view(
                contig=[contig_of[i] for i in idxs],
                rcontig=[-1] * n,
                start=[int(dnms[i]["start"]) for i in idxs],
                end=[int(dnms[i]["end"]) for i in idxs],
                vartype=[vartype_code(dnms[i]["vartype"]) for i in idxs],
                refs=[b""] * n,
                alts=[b""] * n,
                cutoff=0.0,
                mult=[mult_of[i] for i in idxs],
            )
            if not fetch and not attach:
                continue  # the caller runs its own device stage over the batch (K6): no lists on the host
            co, ci, cf, ho, hi = self.backend.find(fam, dv, params, mode)
            col, hol = co.tolist(), ho.tolist()
            for k, i in enumerate(idxs):
                found[i] = dict(
                    mult=mult_of[i],
                    cand_idx=ci[col[k] : col[k + 1]],
                    cand_flags=cf[col[k] : col[k + 1]],
                    het_idx=hi[hol[k] : hol[k + 1]],
                )
            if attach:
                pending_attach.append((idxs, ci, cf, hi, col, hol, dad, mom))

        def attach_now():
            # the site dicts the reference leaves on every DNM (:262-343): the columns of the whole batch's sites become Python values in
            # ONE pass each (a fancy index and a tolist per column, not per DNM), then one dict display per site
            for idxs, ci, cf, hi, col, hol, dad, mom in pending_attach:
                cands_all = self._site_dicts(ci, cf, dad, mom, whole_region)
                hets_all = self._site_dicts(hi, None, dad, mom, False)
                for k, i in enumerate(idxs):
                    dn = dnms[i]
                    cands = cands_all[col[k] : col[k + 1]]
                    hets = hets_all[hol[k] : hol[k + 1]]
                    if many:
                        # find_many only creates the keys on first append (:481-482, :540-541)
                        if cands:
                            dn["candidate_sites"] = cands
                        if hets:
                            dn["het_sites"] = hets
                    else:
                        dn["candidate_sites"] = cands  # :341-342
                        dn["het_sites"] = hets
            del pending_attach[:]
        if not defer_attach:
            attach_now()
        ret = [dnms[i] for i in order] + [dnms[i] for i in auto_tail]
        ret_idx = order + auto_tail
        # (defer_attach: the caller runs info["attach"]() itself -- the read stage needs the index lists, not the dicts, so a staged batch builds
        # them beside the native decode of its first chunks instead of in front of it)
        return ret, {"order": ret_idx, "found": found, "many": many, "mode": mode, "scanned": scan, "contig_of": contig_of, "attach": attach_now}

    def _site_dicts(self, idx, flags, dad, mom, with_kid_allele):
        s = self.sites
        if len(idx) == 0:
            return []
        # columns of the few sites as Python values in one go (tolist), then one dict display per site
        pos = s.pos[idx].tolist()
        ref = _CHR_OBJ[s.ref_base[idx]].tolist()
        alt = _CHR_OBJ[s.alt_base[idx]].tolist()
        if flags is None:
            return [{"pos": p, "ref_allele": r, "alt_allele": a} for p, r, a in zip(pos, ref, alt)]
        out = []
        for p, r, a, fl in zip(pos, ref, alt, flags.tolist()):
            d = {"pos": p, "ref_allele": r, "alt_allele": a}
            if with_kid_allele:
                d["kid_allele"] = ("ref_parent", "alt_parent")[((fl >> abi.CF_KA_SHIFT) & 3) - 1]
            if fl & abi.CF_ALT_DAD:
                d["alt_parent"], d["ref_parent"] = dad, mom
            else:
                d["alt_parent"], d["ref_parent"] = mom, dad
            out.append(d)
        return out

    # find_many bookkeeping (create_lookups :347-396 and the flattening :647-661)
    def _find_many_order(self, dnms, pedigrees, build):
        auto, non = [], []
        for i, dn in enumerate(dnms):
            (auto if autophaseable(dn, pedigrees, build) else non).append(i)
        vbs: Dict[str, Dict[str, Dict[int, List[int]]]] = {}
        for i in non:
            dn = dnms[i]
            vbs.setdefault(dn["kid"], {}).setdefault(dn["chrom"], {}).setdefault(int(dn["start"]), []).append(i)
        order = []
        for kid in vbs:
            for chrom in vbs[kid]:
                for pos in vbs[kid][chrom]:
                    order.extend(vbs[kid][chrom][pos])
        return order, auto

    def _find_many_multiplicity(self, dnms, order):
        """How many times find_many appends each site to a DNM's lists: once per
        occurrence of the kid in samples_by_location[chrom][start] (:385-395, :406-410,
        :454): DNMs of the kid starting there plus DNMs of the kid ending there with
        end - start > 2 (quirk Q6)."""
        sbl: Dict[tuple, Dict[str, int]] = {}
        for i in order:
            dn = dnms[i]
            st, en = int(dn["start"]), int(dn["end"])
            k = (dn["chrom"], st)
            sbl.setdefault(k, {})
            sbl[k][dn["kid"]] = sbl[k].get(dn["kid"], 0) + 1
            if (en - st) > 2:
                k = (dn["chrom"], en)
                sbl.setdefault(k, {})
                sbl[k][dn["kid"]] = sbl[k].get(dn["kid"], 0) + 1
        out = {}
        for i in order:
            dn = dnms[i]
            m = sbl[(dn["chrom"], int(dn["start"]))][dn["kid"]]
            if m > 255:
                raise OverflowError("more than 255 DNMs of one kid share a start position")
            out[i] = m
        return out

    def _find_many_keyerror_chroms(self, dnms, order, threads):
        """find_many with whole_region=True indexes vars_by_sample[...][dn_loc] with
        the END keys as well (:412-415): KeyError on the first variant of the chromosome
        unless every end key is also a start of the same kid (quirk Q7).  With
        threads != 1 the worker's exception is swallowed and the chromosome yields
        nothing; with threads == 1 the reference crashes -- mirrored as KeyError."""
        starts = set()
        for i in order:
            dn = dnms[i]
            starts.add((dn["kid"], dn["chrom"], int(dn["start"])))
        dead = set()
        for i in order:
            dn = dnms[i]
            st, en = int(dn["start"]), int(dn["end"])
            if (en - st) > 2 and (dn["kid"], dn["chrom"], en) not in starts:
                dead.add(dn["chrom"])
        if dead and threads == 1:
            raise KeyError("find_many(whole_region=True): reference raises KeyError (SURVEY.md quirk Q7)")
        return dead

    # ------------------------------------------------------ read phasing
    def get_refalt(self, chrom: str, pos: int):
        """reference snv_phaser.py:73-84: REF of the first and ALTs of all records
        overlapping 1-based [pos, pos+1]."""
        name = self.prefix + chrom.strip("chr")
        idx = self.sites.query(name, int(pos), int(pos) + 1)
        ref = None
        alts: List[str] = []
        for j in idx:
            if ref is None:
                ref = self.sites.ref_str[int(j)]
            alts.extend(self.sites.alt_strs[int(j)])
        return ref, alts

    def batch_refalt(self, dnms, idxs):
        """get_refalt for many DNMs at once: the look-ups (two binary searches per DNM in the reference's per-DNM VCF query) as
        vectorised searches per contig; the strings only for the one or two records each DNM really overlaps.  -> {i: (ref, alts)}"""
        s = self.sites
        out = {}
        by_contig = {}
        for i in idxs:
            name = self.prefix + dnms[i]["chrom"].strip("chr")
            c = s.contig_index.get(name, -1)
            if c < 0:
                out[i] = (None, [])
            else:
                by_contig.setdefault(c, []).append(i)
        lim = np.iinfo(s.pos.dtype)
        for c, members in by_contig.items():
            lo, hi = int(s.contig_off[c]), int(s.contig_off[c + 1])
            seg = s.pos[lo:hi]
            st = np.fromiter((int(dnms[i]["start"]) for i in members), np.int64, len(members))
            span = s._max_span(c)
            # 1-based [pos, pos + 1]: records with start + 1 <= pos + 1 and end >= pos
            k_lo = lo + np.searchsorted(seg, np.clip(st - span, lim.min, lim.max).astype(s.pos.dtype), side="left")
            k_hi = lo + np.searchsorted(seg, np.clip(st, lim.min, lim.max).astype(s.pos.dtype), side="right")
            # the usual case -- exactly one record in the range, overlapping, a plain SNV (not UZ_SF_COMPLEX: one REF base, one one-base ALT) --
            # takes its two strings from the base columns, for all such DNMs at once; everything else asks the table for the strings
            one = (k_hi - k_lo) == 1
            j1 = np.where(one, k_lo, lo).astype(np.int64)
            one &= (s.end[j1] >= st) & ((s.sflags[j1] & SFLAG_COMPLEX) == 0) & (s.ref_base[j1] != 0) & (s.alt_base[j1] != 0)
            r1 = _CHR_OBJ[s.ref_base[j1]].tolist()
            a1 = _CHR_OBJ[s.alt_base[j1]].tolist()
            for i, a, b, p, fast, r, al in zip(members, k_lo.tolist(), k_hi.tolist(), st.tolist(), one.tolist(), r1, a1):
                if fast:
                    out[i] = (r, [al])
                    continue
                ref, alts = None, []
                for j in range(a, b):
                    if s.end[j] >= p:
                        if ref is None:
                            ref = s.ref_str[j]
                        alts.extend(s.alt_strs[j])
                out[i] = (ref, alts)
        return out

    def kid_cutoff(self, kid: str, bam: str, readlen: int, stdevs: int, insert_size_max_sample: int) -> float:
        if kid not in self.cutoffs or not self.cutoffs[kid]:  # snv_phaser.py:133-135, read_collector.py:377
            rt = self.reads_header(bam)
            head = getattr(rt, "tlen_head", None)
            if head is None:
                head = rt.tlen
            # the first insert_size_max_sample + 1 records of the file (read_collector.py:13-17)
            self.cutoffs[kid] = concordant_cutoff(head[: int(insert_size_max_sample) + 1], readlen, stdevs)
        return self.cutoffs[kid]

    def resolve_reads_contig(self, rt: ReadsTable, chrom: str):
        """reference read_collector.py:384-392: fetch on the literal name, on
        ValueError retry with the chr prefix toggled and the narrower window."""
        if chrom in rt.contig_index:
            return rt.contig_index[chrom], 0
        alt = chrom.strip("chr") if "chr" in chrom else "chr" + chrom
        if alt in rt.contig_index:
            return rt.contig_index[alt], abi.DF_FETCH_FALLBACK
        return -1, 0

    def _fetches_of(self, idxs, dnms, prep, found, params, cutoff):
        """the fetches the read stage will make for these DNMs (staging.fetch_points): the DNM position and every het site of its window
        (read_collector.py:385, :167) -> (contig, lo, hi, extra, the batch holds SVs)"""
        from .staging import fetch_points
        het_off = np.zeros(len(idxs) + 1, np.int64)
        for k, i in enumerate(idxs):
            het_off[k + 1] = het_off[k] + len(found[i]["het_idx"])
        het_idx = np.concatenate([found[i]["het_idx"] for i in idxs] + [np.zeros(0, np.int32)]).astype(np.int64)
        vts = [vartype_code(dnms[i]["vartype"]) for i in idxs]
        fc, flo, fhi, fex = fetch_points(
            [prep[i]["tid"] for i in idxs], [int(dnms[i]["start"]) for i in idxs], [prep[i]["dflags"] for i in idxs],
            self.sites.pos, het_off, het_idx, params, vartype=vts,
            end=[int(dnms[i]["end"]) for i in idxs], cutoff=cutoff,
            allele_len=[max(len(prep[i]["ref"]), len(prep[i]["alt"])) for i in idxs])
        # (the read stage asks for the quality bits of "good" records only -- the DNM reads of a point variant and the records
        # registered at het sites, both under goodread, read_collector.py:43-46; SV evidence is collected under
        # goodread(read, True) from flags, CIGARs and mates alone, :476-596 -- so every batch travels with the qualities as
        # counts + short lists and, of the rows, the 32-base units that hold a one-base fetch; the +-cutoff fetches around an
        # SV's breakpoints stage no unit)
        return fc, flo, fhi, fex, any(v != abi.VT_POINT for v in vts)

    def _dnms_view_of(self, idxs, dnms, prep, found, cutoff):
        name_of = {}
        for i in idxs:
            nm = self.prefix + dnms[i]["chrom"].strip("chr")
            name_of[i] = self.sites.contig_index.get(nm, -1)
        return abi.dnms_view(
            contig=[name_of[i] for i in idxs],
            rcontig=[prep[i]["tid"] for i in idxs],
            start=[int(dnms[i]["start"]) for i in idxs],
            end=[int(dnms[i]["end"]) for i in idxs],
            vartype=[vartype_code(dnms[i]["vartype"]) for i in idxs],
            refs=[prep[i]["ref"] for i in idxs],
            alts=[prep[i]["alt"] for i in idxs],
            cutoff=cutoff,
            dflags=[prep[i]["dflags"] for i in idxs],
            mult=[found[i]["mult"] for i in idxs],
        )

    CHUNK_DNMS = 3400  # DNMs per chunk of a large batch (bench.py's files -> results pass: 1500 / 2500 / 3400 / 4000 / 5000 = 23.7 / 24.6 / 25.9 / 24.3 / 22.5 k DNMs/s)

    def _chunked_batch(self, batch, dnms, pedigrees, prep, found, params, readlen, stdevs, insert_size_max_sample, want_lists, mode, results,
                       while_first_stages=None) -> bool:
        """A large batch of ONE kid from an indexed BAM through the staged pipeline (pipeline.run_pipelined: what bench.py measures) -- the
        reference runs one task per DNM on a thread pool, each opening the alignment file for its own window (snv_phaser.py:244-298); here chunk
        k + 1 is decoded on a worker thread (BAM blocks through the index straight into the link form, the blocks inflated on the device) while
        the read stage of chunk k runs, and the vote lists of a chunk are fetched when its read stage is collected.  Fills `results` and returns
        True when it took the batch."""
        import os
        if len(batch) != 1 or not hasattr(self.backend, "stage_reads") or os.environ.get("UZ_HOST_CHUNKS", "1") == "0":
            return False
        (kid, bam), idxs = next(iter(batch.items()))
        stager = getattr(self.reads_by_bam, "stager", None)
        chunk = int(os.environ.get("UZ_HOST_CHUNK_DNMS", self.CHUNK_DNMS))
        if len(idxs) < 2 * chunk or not self._indexed(bam) or stager is None or stager(bam) is None:
            return False
        from concurrent.futures import ThreadPoolExecutor
        src = stager(bam)
        fam = self.family(kid, pedigrees[kid]["dad"], pedigrees[kid]["mom"])
        cutoff = self.kid_cutoff(kid, bam, readlen, stdevs, insert_size_max_sample)
        # The first chunk half-size (nothing hides its decode), the rest in equal chunks of at most `chunk` DNMs (a short tail is spread over them,
        # not joined to the last one: the read stage of a larger last chunk is what nothing hides).  (Round 5 needed equal chunks for another reason --
        # a device slot that met a larger batch grew through hipFree + hipMalloc with every stream waiting; the slots are chosen best-fit and grow
        # without a free since round 6: abi.hip uz_bam_walk, tests/test_bamjoin_gpu.py::test_slots_grow_once.)
        n_all, first = len(idxs), min(len(idxs), chunk // 2)
        k_rest = max(1, -(-(n_all - first) // chunk))
        size = -(-(n_all - first) // k_rest)
        cuts = [0] + [min(n_all, first + j * size) for j in range(k_rest)] + [n_all]
        cuts = sorted(set(cuts))
        parts = [idxs[cuts[k]: cuts[k + 1]] for k in range(len(cuts) - 1)]

        from . import pipeline
        # (chunks of a few thousand DNMs: the first read stage should start as early as it can.  The lag is fixed HERE, UZ_PIPE_LAG is not
        # honoured on this route: the staging slots below are counted from it, and a slot re-staged while the upload of chunk k - 2 still
        # reads its page-locked buffers would corrupt that chunk's records silently)
        lag = 1
        # BAM stages in flight beside the chunk on the device (native code: the interpreter lock is released; bench.py's own loop runs three).
        # Held to the device's walk slots (csrc/uz_ctx.hpp WALK_SLOTS = 4): chunk k still owns one while stage k + ahead asks for another
        ahead = max(1, min(int(os.environ.get("UZ_HOST_AHEAD", 3)), 3))
        n_slots = lag + 1 + ahead  # a slot is staged into again once the read stage of the chunk it held has been collected

        # A chunk's fetch list is made HERE, on the calling thread, before its stage is handed to a worker: the list is a few milliseconds of Python
        # and numpy, and on a worker it waited for the interpreter lock the main thread holds while it builds site dicts and evidence strings
        # (0.08 s of waiting per call for 0.016 s of work, `UZ_HOST_TRACE=1`); the workers are left with native code that needs no lock.
        def fetches(k):
            with _Sec("fetches_of"):
                return self._fetches_of(parts[k], dnms, prep, found, params, cutoff)

        def stage(k, f):
            fc, flo, fhi, fex, has_sv = f
            with _Sec("stage_reads"):
                return self.backend.stage_reads(src, fc, flo, fhi, fex, int(params.min_gt_qual), all_bases=bool(params.no_extended), wide_no_units=has_sv, slot=k % n_slots)

        with ThreadPoolExecutor(ahead) as ex:
            futs = {}
            for k in range(min(ahead, len(parts))):
                futs[k] = ex.submit(stage, k, fetches(k))
            names = {}

            def records(k, het_off, het_idx):  # (the pipeline's own find has just returned: the fetches were derived from the batch's find already)
                packed = futs.pop(k).result()
                if k + ahead < len(parts):
                    futs[k + ahead] = ex.submit(stage, k + ahead, fetches(k + ahead))  # decoded beside the device work of this chunk and the next
                names[k] = packed  # (its `.qnames`: at once for the link form, once the table is built for a batch walked on the device)
                return packed

            def done(k, rr):
              with _Sec("done"):
                part = parts[k]
                lists = None
                if want_lists:
                    with _Sec("votes"):
                        vo, vv = self.backend.votes(len(part))
                    lists = _VoteLists(vo, vv)
                res = dict(status=rr["status"], counts=rr["counts"], origin=rr["origin"], evidence=rr["evidence"], lists=lists)
                table = type("StagedNames", (), {})()
                table.qnames = names.pop(k).qnames
                if lists is not None:
                    # the name and position lists of the chunk's records, built HERE -- beside the native decode of the chunks that follow -- and
                    # not in a pass over the whole batch behind the pipeline
                    with _Sec("evidence"):
                        res["built"] = self._evidence_lists(res, range(len(part)), table)
                # The strings exist (or nobody asked for lists): the chunk's names are let go.  They resolve ids lazily out of the staging slot's
                # page-locked buffers -- kept, they would pin every chunk's stage plan until the call returns, and once the slot is staged into
                # again they would answer with another chunk's names.
                table.qnames = None
                for j, i in enumerate(part):
                    results[i] = (res, j, table)

            if while_first_stages is not None:
                with _Sec("attach"):
                    while_first_stages()  # host work that nothing below waits for, beside the native decode of the first chunks
                    # (on a thread of its own it gains nothing: the chunk pipeline is bound by the device's chain since the joins run there, and the
                    # interpreter lock only moves the waiting around -- measured, round 6)
            chunks = [dict(a=cuts[k], b=cuts[k + 1], dnms=self._dnms_view_of(parts[k], dnms, prep, found, cutoff), records=records, sites=None)
                      for k in range(len(parts))]
            trace = [] if os.environ.get("UZ_HOST_TRACE") else None  # development aid: ms per pipeline step (find, collect, queue, records + upload)
            pipeline.run_pipelined(self.backend, params, mode, len(idxs), chunks, fid=fam, lag=lag, on_done=done, trace=trace, env_lag=False)
            if trace:
                print("[uz] chunked batch, ms per step:", trace[0], file=sys.stderr)
        return True

    def run_read_phasing(self, dnms, pedigrees, *args, **kwargs):
        """run_read_phasing_impl, and whatever it raises behind find() -- a reference KeyError mimicked in pass 1, an I/O error of a BAM stage --
        the caller's DNMs carry their `candidate_sites` / `het_sites`: the reference annotates in find(), before any phasing work starts
        (informative_site_finder.py:341-343), and find(defer_attach=True) only postpones the dict building behind the first BAM stages."""
        box = []
        try:
            return self.run_read_phasing_impl(dnms, pedigrees, *args, _attach_box=box, **kwargs)
        finally:
            for attach in box:
                attach()  # (idempotent: a no-op once the lists are attached)

    def run_read_phasing_impl(
        self,
        dnms,
        pedigrees,
        threads,
        build,
        no_extended,
        multithread_proc_min,
        quiet_mode,
        params: abi.Params,
        search_dist,
        insert_size_max_sample,
        stdevs,
        readlen,
        want_lists=True,
        sv=False,
        _attach_box=None,
    ):
        """reference snv_phaser.py:206-299 (+ multithread_read_phasing :87-203); with sv=True the
        SV variant sv_phaser.py:176-266 (+ :88-173): no REF/ALT lookup, reads collected by
        collect_reads_sv, and autophase that does not short-circuit (quirk Q18)."""
        log = _Log(quiet_mode)
        self._indexed_memo = {}
        params.no_extended = 1 if no_extended else 0
        params.read_goal = int(insert_size_max_sample)
        params.readlen = int(readlen)
        with _Sec("find"):
            ret, info = self.find(
                dnms, pedigrees, search_dist, threads, build, multithread_proc_min, quiet_mode, params,
                whole_region=False, defer_attach=True,
            )
        records: Dict[str, dict] = {}
        if ret is None:
            return records
        attach_sites = info["attach"]
        if _attach_box is not None:
            _attach_box.append(attach_sites)
        found = info["found"]
        # pass 1: host-side filters in the reference's order; collect the device batch per kid
        plan = []  # (dnm index, action)
        batch: Dict[tuple, List[int]] = {}
        prep: Dict[int, dict] = {}
        sample_set = set(self.sites.samples)
        contig_memo: Dict[tuple, tuple] = {}
        with _Sec("refalt"):
            refalt = {} if sv else self.batch_refalt(dnms, [i for i in info["order"] if found.get(i) is not None and len(found[i]["cand_idx"])])
        sec_pass1 = _Sec("pass1").__enter__()
        for i in info["order"]:
            dn = dnms[i]
            dad_id, mom_id = pedigrees[dn["kid"]]["dad"], pedigrees[dn["kid"]]["mom"]
            if autophaseable(dn, pedigrees, build):  # snv_phaser.py:251, :302-352
                plan.append((i, "auto"))
                if not sv:
                    continue
                # sv_phaser.autophase writes the record but returns None (sv_phaser.py:353-354): fall through
            f = found.get(i)
            if f is None or len(f["cand_idx"]) == 0:  # :254-262
                plan.append((i, "nocand"))
                continue
            if dn["kid"] not in sample_set:  # :109-110
                plan.append((i, "silent"))
                continue
            if sv:
                ref, alts = "", [""]
            else:
                ref, alts = refalt[i] if i in refalt else self.get_refalt(dn["chrom"], dn["start"])  # :111-116
                if len(alts) < 1:
                    plan.append((i, "nogt"))
                    continue
                if len(alts) > 1:
                    plan.append((i, "manygt"))
                    continue
            hk = (dn["bam"], dn["chrom"])
            if hk not in contig_memo:
                contig_memo[hk] = self.resolve_reads_contig(self.reads_header(dn["bam"]), dn["chrom"])
            tid, fl = contig_memo[hk]
            if tid < 0:
                plan.append((i, "silent"))  # ValueError out of the worker: no record
                continue
            prep[i] = dict(ref=ref.encode("ascii"), alt=alts[0].encode("ascii"), tid=tid, dflags=fl)
            batch.setdefault((dn["kid"], dn["bam"]), []).append(i)
            plan.append((i, "phase"))
        # pass 2: device.  One batch per kid (its own family columns, alignment file and insert cutoff); several kids go
        # to the device as ONE cohort batch (uz_phase_cohort), not as one launch sequence per kid
        results: Dict[int, dict] = {}
        groups, order_all, tables, handles = [], [], [], []
        sec_pass1.__exit__()
        with _Sec("chunked"):
          chunked = self._chunked_batch(batch, dnms, pedigrees, prep, found, params, readlen, stdevs, insert_size_max_sample, want_lists, info["mode"], results,
                                        while_first_stages=attach_sites)
        attach_sites()  # (a batch that did not take the chunked route: here; otherwise done already, and this is a no-op)
        for (kid, bam), idxs in ([] if chunked else batch.items()):
            dad_id, mom_id = pedigrees[kid]["dad"], pedigrees[kid]["mom"]
            fam = self.family(kid, dad_id, mom_id)
            cutoff = self.kid_cutoff(kid, bam, readlen, stdevs, insert_size_max_sample)
            region_table = None
            if self._indexed(bam):
                # decode only what the batch's fetches can return (+ mates) through the index: the read stage looks at
                # nothing else (read_collector.py:385, :167, :400), so nothing else is inflated, staged or uploaded
                fc, flo, fhi, fex, has_sv = self._fetches_of(idxs, dnms, prep, found, params, cutoff)
                stager = getattr(self.reads_by_bam, "stager", None)
                src = stager(bam) if (stager and hasattr(self.backend, "upload_reads_staged")) else None
                if src is not None:
                    # BAM + BAI -> the link form in one pass (uz_bam_stage_*): no table in between
                    rh, region_table = self.backend.upload_reads_staged(src, fc, flo, fhi, fex, int(params.min_gt_qual),
                                                                        all_bases=bool(params.no_extended), wide_no_units=has_sv)
                else:
                    region_table = self.reads_by_bam.regions(bam, fc, flo, fhi)
                    rh = self.backend.upload_reads(region_table, min_base_qual=int(params.min_gt_qual), point_only=True,
                                                   fetches=(fc, flo, fhi, fex), all_bases=bool(params.no_extended), wide_no_units=has_sv)
                handles.append(rh)
            else:
                rh = self.reads(bam, params.min_gt_qual)
            groups.append((fam, rh, len(order_all), len(idxs), cutoff))
            tables.append(region_table if region_table is not None else self.reads_by_bam[bam])
            order_all.extend(idxs)
        if order_all:
            dv = self._dnms_view_of(order_all, dnms, prep, found, groups[0][4])
            fl = [found[i] for i in order_all]
            if len(groups) == 1:
                res = self.backend.phase(groups[0][0], groups[0][1], dv, params, fl, want_lists, find_mode=info["mode"])
            else:
                res = self.backend.phase_cohort(groups, dv, params, fl, want_lists, find_mode=info["mode"])
            free = getattr(self.backend, "free_reads", None)
            for rh in handles:  # region tables live for one batch
                if free:
                    free(rh)
            for g, (fam, rh, first, count, cutoff) in enumerate(groups):
                for k in range(first, first + count):
                    results[order_all[k]] = (res, k, tables[g])
        self._indexed_memo = None
        # pass 3: records, in the reference's order.  The names of the reads in the result lists: one look-up per table for the whole batch
        # (a staged table answers ids through the C ABI: hundreds of thousands of single calls were a third of round 3's host time); the chunks
        # of a staged batch have built theirs already (res["built"])
        sec_pass3 = _Sec("pass3").__enter__()
        by_res: Dict[tuple, tuple] = {}
        for (res, k, rt) in results.values():
            if "built" not in res and res.get("lists") is not None:
                by_res.setdefault((id(res), id(rt)), (res, rt, []))[2].append(k)  # (a cohort batch: one result, a table per kid)
        for res, rt, ks in by_res.values():
            res.setdefault("built_", {}).update(self._evidence_lists(res, ks, rt))
        for res, rt, ks in by_res.values():
            if "built_" in res:
                res["built"] = res.pop("built_")
        for i, action in plan:
            dn = dnms[i]
            dad_id, mom_id = pedigrees[dn["kid"]]["dad"], pedigrees[dn["kid"]]["mom"]
            region = {"chrom": dn["chrom"], "start": dn["start"], "end": dn["end"]}
            key = "_".join([str(v) for v in region.values()] + [dn["kid"], dn["vartype"]])
            if action == "auto":
                records[key] = {
                    "region": region, "vartype": dn["vartype"], "kid": dn["kid"], "dad": dad_id, "mom": mom_id,
                    "cnv_dad_sites": "NA", "cnv_mom_sites": "NA", "cnv_evidence_type": "SEX-CHROM",
                    "dad_sites": "", "mom_sites": "", "evidence_type": "SEX-CHROM",
                    "dad_reads": [], "mom_reads": [],
                }
            elif action == "nocand":
                if sv:  # sv_phaser.py:227-229
                    log("No usable informative sites for read-based phasing of variant {}:{}-{}".format(
                        dn["chrom"], dn["start"], dn["end"]))
                else:
                    log("No usable informative sites for variant {}:{}-{}".format(dn["chrom"], dn["start"], dn["end"]))
            elif action == "nogt":
                log("No usable genotype for variant {chrom}:{start}-{end}".format(**region))
            elif action == "manygt":
                log("Too many genotypes for variant {chrom}:{start}-{end}".format(**region))
            elif action == "phase":
                res, k, rt = results[i]
                st = int(res["status"][k])
                if st == abi.ST_NO_OVERLAP:
                    log("No reads overlap informative sites for variant {chrom}:{start}-{end}".format(**region))
                    continue
                if st == abi.ST_CAPACITY:
                    # include/uz_types.h: reported, never silently dropped -- printed even under --quiet
                    msg = ("variant {chrom}:{start}-{end} of %s exceeds the device layout of the read stage "
                           "(UZ_ST_CAPACITY): NOT phased" % dn["kid"]).format(**region)
                    print(msg, file=sys.stderr)
                    self.capacity_skipped.append(key)
                    continue
                if st != abi.ST_OK:
                    # ST_REF_EXCEPTION: the reference's worker raises (KeyError in connect_reads) and the default
                    # thread pool swallows it: no record, no message
                    continue
                if res.get("lists") is not None:
                    dad_reads, mom_reads, dad_sites, mom_sites = res["built"][k]
                else:
                    c = res["counts"][k]
                    dad_reads, mom_reads = [None] * int(c[0]), [None] * int(c[1])
                    dad_sites, mom_sites = [None] * int(c[2]), [None] * int(c[3])
                records[key] = {
                    "region": region, "vartype": dn["vartype"], "kid": dn["kid"], "dad": dad_id, "mom": mom_id,
                    "dad_sites": dad_sites, "mom_sites": mom_sites, "evidence_type": "readbacked",
                    "dad_reads": dad_reads, "mom_reads": mom_reads,
                    "cnv_dad_sites": "", "cnv_mom_sites": "", "cnv_evidence_type": "",
                }
        sec_pass3.__exit__()
        _Sec.report()
        return records

    @staticmethod
    def _evidence_lists(res, ks, rt) -> dict:
        """{k: (dad_reads, mom_reads, dad_sites, mom_sites)} of the phased DNMs k of one device result: read names through ONE look-up for all of
        them (a staged table answers ids through the C ABI), positions as the reference's strings (snv_phaser.py:169-185)"""
        lists, status = res["lists"], res["status"]
        st = status.tolist() if hasattr(status, "tolist") else list(status)
        ok = [k for k in ks if st[k] == abi.ST_OK]
        out = {}
        if not ok:
            return out
        take = getattr(getattr(rt, "qnames", None), "take", None)
        fast = isinstance(lists, _VoteLists)
        if fast and take is not None and len(ok) * 2 >= len(lists):  # (a chunk's lists as a whole: its DNMs' lists are slices of one list of strings)
            sl, o = lists.strings(take), lists._vol
            for k in ok:
                out[k] = (sl[o[4 * k]: o[4 * k + 1]], sl[o[4 * k + 1]: o[4 * k + 2]], sl[o[4 * k + 2]: o[4 * k + 3]], sl[o[4 * k + 3]: o[4 * k + 4]])
            return out
        nm = None
        if take is not None:
            if fast:
                u = lists.read_ids(ok)
            else:
                ids = [lists[k][q] for k in ok for q in (0, 1)]
                u = np.unique(np.concatenate(ids + [np.zeros(0, np.int64)]).astype(np.int64))
            nm = dict(zip(u.tolist(), take(u)))
        for k in ok:
            if fast:
                dr, mr, ds, ms = lists.py(k)  # (Python lists already)
            else:
                dr, mr, ds, ms = (x.tolist() for x in lists[k])
            if nm is not None:
                dad_reads = [nm[q] for q in dr]
                mom_reads = [nm[q] for q in mr]
            else:
                dad_reads = [rt.qnames[q] for q in dr]
                mom_reads = [rt.qnames[q] for q in mr]
            out[k] = (dad_reads, mom_reads, list(map(str, ds)), list(map(str, ms)))
        return out

    # ------------------------------------------------------- CNV phasing
    def run_cnv_phasing(self, dnms, pedigrees, threads, build, multithread_proc_min, quiet_mode, params, annotate=True):
        """reference sv_phaser.py:357-423 with phase_by_snvs (:71-85) and
        multithread_cnv_phasing (:269-301): allele-balance phasing of DEL/DUP."""
        log = _Log(quiet_mode)
        many = len(dnms) >= multithread_proc_min
        # find_many's whole-region mode is the quirk path (Q7: KeyError / appended lists): it keeps the annotated site
        # dicts; the per-DNM path (every realistic CNV batch) takes the counts and ordered site lists from K6.
        # annotate: write `candidate_sites` into the DNM dicts as the reference's find does (informative_site_finder.py:
        # 341-343); phase_svs turns it off -- its read-backed find overwrites the lists before anyone can see them
        ret, info = self.find(
            dnms, pedigrees, 0, threads, build, multithread_proc_min, quiet_mode, params, whole_region=True,
            attach=many or annotate, fetch=many or annotate,
        )
        records: Dict[str, dict] = {}
        if ret is None:
            return records
        device_lists: Dict[int, tuple] = {}
        if not many:
            by_kid: Dict[str, List[int]] = {}
            for i in info["scanned"]:
                if dnms[i]["vartype"] in ["DEL", "DUP"]:
                    by_kid.setdefault(dnms[i]["kid"], []).append(i)
            for kid, idxs in by_kid.items():
                fam = self.family(kid, pedigrees[kid]["dad"], pedigrees[kid]["mom"])
                n = len(idxs)
                dv = abi.dnms_view(
                    contig=[info["contig_of"][i] for i in idxs], rcontig=[-1] * n,
                    start=[int(dnms[i]["start"]) for i in idxs], end=[int(dnms[i]["end"]) for i in idxs],
                    vartype=[vartype_code(dnms[i]["vartype"]) for i in idxs], refs=[b""] * n, alts=[b""] * n, cutoff=0.0,
                )
                res = self.backend.phase_cnv(fam, dv, params)
                for k, i in enumerate(idxs):
                    device_lists[i] = res["lists"][k]
        for i in info["order"]:
            dn = dnms[i]
            dad_id, mom_id = pedigrees[dn["kid"]]["dad"], pedigrees[dn["kid"]]["mom"]
            # sv_phaser.autophase writes the record but returns None (quirk Q18): fall through
            region = {"chrom": dn["chrom"], "start": dn["start"], "end": dn["end"]}
            key = "_".join([str(r) for r in region.values()] + [dn["kid"], dn["vartype"]])
            if autophaseable(dn, pedigrees, build):
                records[key] = {
                    "region": region, "vartype": dn["vartype"], "kid": dn["kid"], "dad": dad_id, "mom": mom_id,
                    "cnv_dad_sites": "NA", "cnv_mom_sites": "NA", "cnv_evidence_type": "SEX-CHROM",
                    "dad_sites": "", "mom_sites": "", "evidence_type": "SEX-CHROM",
                    "dad_reads": [], "mom_reads": [],
                }
            if dn["vartype"] not in ["DEL", "DUP"]:  # :401
                continue
            if many:
                cs = dn.get("candidate_sites") or []
                ev = {dad_id: [], mom_id: []}
                if cs:
                    origin = {cs[0]["ref_parent"]: [], cs[0]["alt_parent"]: []}  # :75-78
                    for s in cs:
                        origin[s[s["kid_allele"]]].append(s)  # :81-84
                    for parent in ev:
                        if parent in origin and len(origin[parent]) > 0:
                            ev[parent] = [str(o["pos"]) for o in origin[parent]]
                n_cand = len(cs)
                dad_sites, mom_sites = ev[dad_id], ev[mom_id]
            else:
                dl = device_lists.get(i)
                dad_sites = [str(int(p)) for p in dl[0]] if dl is not None else []
                mom_sites = [str(int(p)) for p in dl[1]] if dl is not None else []
                n_cand = len(dad_sites) + len(mom_sites)
            if n_cand == 0:  # :404-412
                log("No usable informative sites for allele-balance phasing of variant {}:{}-{}".format(
                    dn["chrom"], dn["start"], dn["end"]))
                continue
            records[key] = {
                "region": region, "vartype": dn["vartype"], "kid": dn["kid"], "dad": dad_id, "mom": mom_id,
                "cnv_dad_sites": dad_sites, "cnv_mom_sites": mom_sites, "cnv_evidence_type": "ALLELE-BALANCE",
                "dad_sites": "", "mom_sites": "", "evidence_type": "",
                "dad_reads": [], "mom_reads": [],
            }
        return records
