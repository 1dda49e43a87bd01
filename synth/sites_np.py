"""Vectorised (numpy) synthetic sites table + DNM list at benchmark scale.

Test / bench infrastructure.  Produces the device-facing columns directly
(no record objects): a whole-genome-like sorted sites table for one trio and a
list of DNM coordinates placed on existing sites (kid het, parents hom-ref), as
SURVEY.md 8(d) configs 2/3 describe.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

GRCH38_LEN = [
    248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
    138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
    83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415,
]


@dataclass
class SitesColumns:
    contig_names: List[str]
    contig_off: np.ndarray  # int64 [nc+1]
    pos: np.ndarray  # int32
    sflags: np.ndarray  # uint8
    ref_base: np.ndarray
    alt_base: np.ndarray
    gt: np.ndarray  # uint8 packed kid|dad<<2|mom<<4
    rd: np.ndarray  # uint16 [3][S]
    ad: np.ndarray
    gq: np.ndarray
    khap: np.ndarray = None  # uint8: bit0 allele on the kid's paternal haplotype, bit1 maternal

    @property
    def n(self):
        return int(self.pos.shape[0])


def make_sites(n_sites: int, seed: int = 202, contig_lens=None, complex_frac=0.02, weird_frac=0.01) -> SitesColumns:
    rng = np.random.default_rng(seed)
    lens = np.array(contig_lens if contig_lens is not None else GRCH38_LEN, dtype=np.float64)
    per = np.floor(lens / lens.sum() * n_sites).astype(np.int64)
    per[0] += n_sites - per.sum()
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    off[1:] = np.cumsum(per)
    pos = np.empty(n_sites, dtype=np.int32)
    for c in range(len(lens)):
        k = int(per[c])
        if k == 0:
            continue
        step = lens[c] / k
        base = (np.arange(k) * step).astype(np.int64)
        jit = (rng.random(k) * max(1.0, step - 1)).astype(np.int64)
        pos[off[c]: off[c + 1]] = (base + jit + 1).astype(np.int32)
    S = n_sites
    f = rng.random(S) ** 3
    dh = (rng.random((2, S)) < f).astype(np.uint8)
    mh = (rng.random((2, S)) < f).astype(np.uint8)
    td = rng.integers(0, 2, S)
    tm = rng.integers(0, 2, S)
    pat = np.where(td == 0, dh[0], dh[1])
    mat = np.where(tm == 0, mh[0], mh[1])
    code = np.array([0, 1, 3], dtype=np.uint8)
    g = np.stack([code[pat + mat], code[dh[0] + dh[1]], code[mh[0] + mh[1]]])
    depth = rng.poisson(30.0, (3, S)).astype(np.int64)
    p_alt = np.where(g == 0, 0.01, np.where(g == 1, 0.5, 0.99))
    ad = rng.binomial(depth, p_alt).astype(np.int64)
    rd = depth - ad
    gq = np.where(rng.random((3, S)) < 0.93, 99, rng.integers(0, 61, (3, S))).astype(np.int64)
    # awkward values: unknown genotypes, missing fields, zero depth
    w = rng.random((3, S)) < weird_frac
    kind = rng.integers(0, 4, (3, S))
    g = np.where(w & (kind == 0), 2, g).astype(np.uint8)
    miss = w & (kind == 1)
    rd = np.where(miss, 0xFFFF, rd)
    ad = np.where(miss, 0xFFFF, ad)
    gq = np.where(w & (kind == 2), 0xFFFF, gq)
    zero = w & (kind == 3)
    rd = np.where(zero, 0, rd)
    ad = np.where(zero, 0, ad)
    sflags = (rng.random(S) < complex_frac).astype(np.uint8)
    ref_i = rng.integers(0, 4, S)
    alt_i = (ref_i + rng.integers(1, 4, S)) % 4
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    ref_base = np.where(sflags == 1, 0, bases[ref_i]).astype(np.uint8)
    alt_base = np.where(sflags == 1, 0, bases[alt_i]).astype(np.uint8)
    gt = (g[0] | (g[1] << 2) | (g[2] << 4)).astype(np.uint8)
    names = [str(i + 1) for i in range(22)] + ["X", "Y"]
    names = names[: len(lens)] if len(lens) <= 24 else [str(i + 1) for i in range(len(lens))]
    return SitesColumns(
        names, off, pos, sflags, ref_base, alt_base, gt,
        np.ascontiguousarray(rd.astype(np.uint16)), np.ascontiguousarray(ad.astype(np.uint16)),
        np.ascontiguousarray(gq.astype(np.uint16)),
        khap=(pat | (mat << 1)).astype(np.uint8),
    )


@dataclass
class DnmColumns:
    site_idx: np.ndarray  # int32: the DNM's own record in the sites table
    contig: np.ndarray  # int32
    start: np.ndarray  # int32 (0-based)
    end: np.ndarray
    kind: np.ndarray  # uint8 0 SNV, 1 insertion, 2 deletion
    length: np.ndarray  # uint8 inserted / deleted bases
    origin: np.ndarray  # uint8 0 = paternal haplotype (dad) carries the DNM, 1 = maternal
    refs: list
    alts: list

    @property
    def n(self):
        return int(self.start.shape[0])


def place_dnms_full(sc: SitesColumns, n_dnms: int, seed: int = 201, indel_frac=0.1, half_width=6000, max_cluster_len=80000,
                    uniform=True) -> DnmColumns:
    """The DNM list of SURVEY.md 8(d) config 3: positions UNIFORM over the genome (so the +-search_dist windows of
    neighbours overlap as often as they do for real DNMs), 90 % SNV / 10 % small INDEL, an origin haplotype and the
    REF/ALT strings.  Every DNM sits on a record of the sites table which becomes the DNM's own record (kid het,
    parents hom-ref).  A DNM that would stretch a chain of overlapping read windows beyond max_cluster_len is
    re-drawn (the generator sorts one cluster's records in LDS)."""
    rng = np.random.default_rng(seed)
    S = sc.n
    contig_of = (np.searchsorted(sc.contig_off, np.arange(S), side="right") - 1).astype(np.int32)
    ok = sc.pos.astype(np.int64) - half_width - 64 > 0  # no read starts left of the contig
    cand = np.nonzero(ok)[0]
    if uniform:
        idx = np.sort(rng.choice(cand, size=n_dnms, replace=False))
    else:  # stratified (no overlapping windows when the table is sparse enough)
        step = cand.size / n_dnms
        idx = cand[np.minimum((np.arange(n_dnms) * step + rng.random(n_dnms) * step * 0.5).astype(np.int64), cand.size - 1)]
    for _ in range(50):  # thin out chains of overlapping windows that would grow beyond max_cluster_len
        pos = sc.pos[idx].astype(np.int64)
        cg = contig_of[idx]
        brk = np.ones(idx.size, bool)
        brk[1:] = (cg[1:] != cg[:-1]) | (pos[1:] - half_width >= pos[:-1] + half_width)
        cid = np.cumsum(brk) - 1
        lo = np.minimum.reduceat(pos, np.nonzero(brk)[0]) - half_width
        hi = np.maximum.reduceat(pos, np.nonzero(brk)[0]) + half_width
        too_long = (hi - lo) > max_cluster_len
        if not too_long.any():
            break
        # drop the last DNM of every over-long cluster and draw replacements elsewhere
        last_of = np.zeros(idx.size, bool)
        last_of[np.r_[np.nonzero(brk)[0][1:] - 1, idx.size - 1]] = True
        drop = last_of & too_long[cid]
        keep = idx[~drop]
        pool = np.setdiff1d(cand, keep, assume_unique=False)
        idx = np.sort(np.concatenate([keep, rng.choice(pool, size=int(drop.sum()), replace=False)]))
    else:
        raise ValueError("could not place %d DNMs with clusters <= %d bp" % (n_dnms, max_cluster_len))
    n = n_dnms
    sc.gt[idx] = 1  # kid het, dad/mom hom-ref
    sc.sflags[idx] = 0
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    sc.ref_base[idx] = bases[rng.integers(0, 4, n)]
    sc.alt_base[idx] = bases[(np.searchsorted(bases, sc.ref_base[idx]) + 1) % 4]
    for m in range(3):
        sc.rd[m][idx] = 30 if m else 15
        sc.ad[m][idx] = 0 if m else 15
        sc.gq[m][idx] = 99
    sc.khap[idx] = 0
    contig = contig_of[idx].astype(np.int32)
    start = sc.pos[idx].astype(np.int32)
    is_indel = rng.random(n) < indel_frac
    kind = np.where(is_indel, rng.integers(1, 3, n), 0).astype(np.uint8)
    length = np.where(is_indel, rng.integers(1, 11, n), 0).astype(np.uint8)
    origin = rng.integers(0, 2, n).astype(np.uint8)
    end = (start + 1 + np.where(kind == 2, length, 0)).astype(np.int32)
    refs, alts = [], []
    for i in range(n):
        r = bytes([sc.ref_base[idx[i]]])
        a = bytes([sc.alt_base[idx[i]]])
        if kind[i] == 1:
            a = r + b"A" * int(length[i])
        elif kind[i] == 2:
            r = r + b"C" * int(length[i])
            a = r[:1]
        refs.append(r)
        alts.append(a)
    return DnmColumns(idx.astype(np.int32), contig, start, end, kind, length, origin, refs, alts)


@dataclass
class Clusters:
    """Merged read windows of neighbouring DNMs: one pile-up each (synth/uzsynth.h)."""
    contig: np.ndarray  # int32
    lo: np.ndarray  # int32 reads are laid out within [lo, hi)
    hi: np.ndarray
    d0: np.ndarray  # int32 DNMs [d0, d0 + nd)
    nd: np.ndarray
    pair_off: np.ndarray  # int64 [n+1]

    @property
    def n(self):
        return int(self.lo.shape[0])

    def of_dnm(self, d):
        """cluster holding DNM d"""
        return int(np.searchsorted(self.d0, d, side="right") - 1)


def make_clusters(dn: DnmColumns, half_width=6000, pairs_per_bp=0.1) -> Clusters:
    pos = dn.start.astype(np.int64)
    brk = np.ones(dn.n, bool)
    brk[1:] = (dn.contig[1:] != dn.contig[:-1]) | (pos[1:] - half_width >= pos[:-1] + half_width)
    first = np.nonzero(brk)[0]
    lo = np.minimum.reduceat(pos, first) - half_width
    hi = np.maximum.reduceat(pos, first) + half_width
    nd = np.diff(np.r_[first, dn.n])
    per_hap = np.maximum(1, np.round((hi - lo - 900) * pairs_per_bp / 2).astype(np.int64))
    pair_off = np.zeros(first.size + 1, np.int64)
    pair_off[1:] = np.cumsum(2 * per_hap)
    return Clusters(dn.contig[first].astype(np.int32), lo.astype(np.int32), hi.astype(np.int32), first.astype(np.int32),
                    nd.astype(np.int32), pair_off)


def place_dnms(sc: SitesColumns, n_dnms: int, seed: int = 201, indel_frac=0.1):
    """Pick n_dnms sites (stratified over the table) and turn them into the DNMs' own
    records: kid het, parents hom-ref, good depth/GQ.  Returns (site index, contig id,
    start, end) arrays; `end - start` > 1 marks a small deletion."""
    rng = np.random.default_rng(seed)
    S = sc.n
    step = S / n_dnms
    idx = (np.arange(n_dnms) * step + rng.random(n_dnms) * step * 0.5).astype(np.int64)
    idx = np.minimum(idx, S - 1)
    sc.gt[idx] = 1  # kid het, dad/mom hom-ref
    sc.sflags[idx] = 0
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    sc.ref_base[idx] = bases[rng.integers(0, 4, n_dnms)]
    sc.alt_base[idx] = bases[(np.searchsorted(bases, sc.ref_base[idx]) + 1) % 4]
    for m in range(3):
        sc.rd[m][idx] = 30 if m else 15
        sc.ad[m][idx] = 0 if m else 15
        sc.gq[m][idx] = 99
    contig = (np.searchsorted(sc.contig_off, idx, side="right") - 1).astype(np.int32)
    start = sc.pos[idx].astype(np.int32)
    dlen = np.where(rng.random(n_dnms) < indel_frac, rng.integers(2, 11, n_dnms), 1).astype(np.int32)
    return idx, contig, start, start + dlen


@dataclass
class CnvColumns:
    """DEL / DUP events for the allele-balance path."""
    contig: np.ndarray  # int32
    start: np.ndarray  # int32 0-based
    end: np.ndarray
    vartype: np.ndarray  # uint8 1 DEL, 2 DUP (UZ_VT_*)
    origin: np.ndarray  # uint8 0 = the event is on the paternal chromosome (dad), 1 = maternal

    @property
    def n(self):
        return int(self.start.shape[0])


def place_cnvs(sc: SitesColumns, n_events: int, seed: int = 501, redraw_seed: int = 502, min_len=1000, max_len=300000) -> CnvColumns:
    """n_events DEL / DUP (50 / 50), length log-uniform in [min_len, max_len], stratified over the genome so that events
    do not overlap; the kid's genotype / depths at the interior sites are re-drawn for a hemizygous kid (DEL: only the
    other parent's allele, half the depth) or a 2:1 allele balance (DUP: two copies of the origin parent's allele, 1.5x
    depth).  Modifies sc.gt / sc.rd[0] / sc.ad[0] in place."""
    rng = np.random.default_rng(seed)
    nc = len(sc.contig_off) - 1
    clen = np.array([int(sc.pos[sc.contig_off[c + 1] - 1]) if sc.contig_off[c + 1] > sc.contig_off[c] else 0 for c in range(nc)], np.int64)
    per = np.floor(clen / clen.sum() * n_events).astype(np.int64)
    per[np.argmax(clen)] += n_events - per.sum()
    contig, start, end = [], [], []
    for c in range(nc):
        k = int(per[c])
        if k == 0:
            continue
        cell = clen[c] / k
        ln = np.exp(rng.uniform(np.log(min_len), np.log(max_len), k)).astype(np.int64)
        ln = np.minimum(ln, max(int(cell) - 2000, min_len))
        st = (np.arange(k) * cell + 1000 + rng.random(k) * np.maximum(cell - ln - 2000, 1)).astype(np.int64)
        contig.append(np.full(k, c, np.int32)); start.append(st); end.append(st + ln)
    contig = np.concatenate(contig); start = np.concatenate(start).astype(np.int32); end = np.concatenate(end).astype(np.int32)
    n = contig.size
    vartype = np.where(rng.random(n) < 0.5, 1, 2).astype(np.uint8)
    origin = rng.integers(0, 2, n).astype(np.uint8)
    # interior sites: 1-based POS in [start, end]  <=>  0-based pos in [start - 1, end - 1]
    r2 = np.random.default_rng(redraw_seed)
    a = np.empty(n, np.int64); b = np.empty(n, np.int64)
    for c in range(nc):
        m = contig == c
        seg = sc.pos[sc.contig_off[c]: sc.contig_off[c + 1]]
        a[m] = sc.contig_off[c] + np.searchsorted(seg, start[m] - 1, side="left")
        b[m] = sc.contig_off[c] + np.searchsorted(seg, end[m] - 1, side="right")
    cnt = (b - a).astype(np.int64)
    ev = np.repeat(np.arange(n), cnt)
    si = (np.repeat(a, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))).astype(np.int64)
    pat, mat = (sc.khap[si] & 1).astype(np.int64), ((sc.khap[si] >> 1) & 1).astype(np.int64)
    o = origin[ev].astype(np.int64)
    is_del = vartype[ev] == 1
    kept = np.where(o == 0, mat, pat)           # DEL: the allele of the chromosome that is left
    dup, oth = np.where(o == 0, pat, mat), np.where(o == 0, mat, pat)
    n_alt = np.where(is_del, kept, 2 * dup + oth)
    copies = np.where(is_del, 1, 3)
    gt_kid = np.where(n_alt == 0, 0, np.where(n_alt == copies, 3, 1)).astype(np.uint8)
    depth = r2.poisson(np.where(is_del, 15.0, 45.0))
    p_alt = np.clip(n_alt / copies, 0.01, 0.99)
    ad = r2.binomial(depth, p_alt)
    real = sc.sflags[si] == 0
    si, gt_kid, ad, depth = si[real], gt_kid[real], ad[real], depth[real]
    sc.gt[si] = (sc.gt[si] & 0xFC) | gt_kid
    sc.ad[0][si] = ad.astype(np.uint16)
    sc.rd[0][si] = (depth - ad).astype(np.uint16)
    return CnvColumns(contig, start, end, vartype, origin)


def breakpoint_dnms(cv: CnvColumns) -> DnmColumns:
    """The breakpoints of the events as a sorted pseudo-DNM list for the read generator (no site of their own: plain
    30x pile-ups around both ends, i.e. no split / discordant support for the read-backed branch)."""
    contig = np.concatenate([cv.contig, cv.contig])
    pos = np.concatenate([cv.start, cv.end])
    order = np.lexsort((pos, contig))
    k = order.size
    z8 = np.zeros(k, np.uint8)
    out = DnmColumns(np.full(k, -1, np.int32), contig[order].astype(np.int32), pos[order].astype(np.int32), (pos[order] + 1).astype(np.int32),
                     z8, z8.copy(), z8.copy(), [b""] * k, [b""] * k)
    inv = np.empty(k, np.int64)
    inv[order] = np.arange(k)
    out.bp_start, out.bp_end = inv[: cv.contig.size], inv[cv.contig.size:]  # entry of every event's start / end breakpoint in the list
    return out
