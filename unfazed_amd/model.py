"""Decoded-input model: what the host hands to the C ABI.

Two record types (one VCF record of the sites file, one BAM alignment record)
and the two column-array (SoA) tables built from them.  The tables are the
"decoded once on the host" inputs of the hot path; their columns are exactly the
arrays `include/unfazed_hip.h` takes.

Semantics restated from the libraries the reference calls (SURVEY.md Appendix B;
cyvcf2 0.31.0 / pysam 0.22.1, neither present in /root/reference):

* site ``start`` is ``Variant.start`` (0-based), ``end`` is ``Variant.end``
  (start + rlen) -- reference informative_site_finder.py:263, unfazed.py:73-74.
* genotype codes are cyvcf2 ``gt_types`` with gts012=False: 0 HOM_REF, 1 HET,
  2 UNKNOWN, 3 HOM_ALT (reference utils.py:2-5).
* a missing depth / GQ is -1 (cyvcf2), stored in the 16-bit device columns as
  the sentinel 0xFFFF.
* segment ``end`` is htslib ``bam_endpos`` (start+1 for a record without
  reference-consuming CIGAR ops); ``mate`` is the record
  ``pysam.AlignmentFile.mate`` would return (first record overlapping
  [mpos, mpos+1) on the mate contig with the opposite READ1/READ2 bit and the
  same query name), or -1 where pysam raises ValueError.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

HOM_REF, HET, GT_UNKNOWN, HOM_ALT = 0, 1, 2, 3  # reference utils.py:2-5

U16_MISSING = 0xFFFF
U16_MAX_VALUE = 0x7FFF  # the device reads the 16-bit columns as signed halfwords (0xFFFF = -1)

# BAM flag bits
FPAIRED, FPROPER, FUNMAP, FMUNMAP, FREVERSE, FMREVERSE = 1, 2, 4, 8, 16, 32
FREAD1, FREAD2, FSECONDARY, FQCFAIL, FDUP, FSUPP = 64, 128, 256, 512, 1024, 2048

# CIGAR op codes (BAM): reference utils.py:13-24
CIGAR_OPS = "MIDNSHP=XB"
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X, OP_B = range(10)
_REF_CONSUMING = (OP_M, OP_D, OP_N, OP_EQ, OP_X)

AUX_MATE_SAME_TID = 1
AUX_HAS_SA = 2
AUX_DECODE_BAD = 4  # no CIGAR / no sequence / no qualities: never a "good read"

SFLAG_COMPLEX = 1  # len(ALT)!=1 or len(REF)>1 or '*' in ALT or len(ALT[0])>1


@dataclass
class SiteRecord:
    """One record of the sites VCF, already decoded."""

    chrom: str
    start: int  # 0-based (POS-1)
    ref: str
    alts: List[str]
    gt_types: Sequence[int]
    ref_depths: Sequence[int]
    alt_depths: Sequence[int]
    gt_quals: Sequence[float]
    end: Optional[int] = None  # start + rlen; default start + len(ref)
    info: Dict[str, object] = field(default_factory=dict)
    genotypes: Optional[List[List[object]]] = None  # [[a0, a1, phased], ...]
    # raw text columns kept for the VCF writer (surface only)
    raw: Optional[List[str]] = None

    def __post_init__(self):
        if self.end is None:
            self.end = self.start + len(self.ref)

    @property
    def is_complex(self) -> bool:
        # reference informative_site_finder.py:239-243 / :571-575
        return (
            len(self.alts) != 1
            or len(self.ref) > 1
            or "*" in self.alts
            or len(self.alts[0]) > 1
        )


@dataclass
class Segment:
    """One BAM alignment record, already decoded."""

    qname: str
    flag: int
    tid: int
    pos: int  # reference_start
    mapq: int
    cigar: List[Tuple[int, int]]  # (op, len), BAM op codes
    mtid: int
    mpos: int
    tlen: int
    seq: str
    qual: Optional[Sequence[int]]
    has_sa: bool = False

    @property
    def ref_len(self) -> int:
        return sum(l for op, l in self.cigar if op in _REF_CONSUMING)

    @property
    def endpos(self) -> int:
        """htslib bam_endpos."""
        if self.flag & FUNMAP or not self.cigar:
            return self.pos + 1
        rl = self.ref_len
        return self.pos + (rl if rl > 0 else 1)


def _pad16(n: int) -> int:
    return (n + 15) & ~15


class SitesTable:
    """SoA columns of a decoded sites VCF (all samples), host side.

    Device-facing per-family columns are produced by :meth:`family_columns`.
    """

    def __init__(self, samples: Sequence[str], contigs: Sequence[str]):
        self.samples = list(samples)
        self.contigs = list(contigs)
        self.contig_index = {c: i for i, c in enumerate(self.contigs)}
        self.contig_off = np.zeros(len(self.contigs) + 1, dtype=np.int64)
        self.pos = np.zeros(0, dtype=np.int32)
        self.end = np.zeros(0, dtype=np.int32)
        self.sflags = np.zeros(0, dtype=np.uint8)
        self.ref_base = np.zeros(0, dtype=np.uint8)
        self.alt_base = np.zeros(0, dtype=np.uint8)
        ns = len(self.samples)
        self.gt = np.zeros((ns, 0), dtype=np.uint8)
        self.ref_depth = np.zeros((ns, 0), dtype=np.int32)
        self.alt_depth = np.zeros((ns, 0), dtype=np.int32)
        self.gq = np.zeros((ns, 0), dtype=np.float64)
        # host-only allele strings (needed by get_refalt, reference snv_phaser.py:73-84)
        self.ref_str: List[str] = []
        self.alt_strs: List[List[str]] = []

    @property
    def n_sites(self) -> int:
        return int(self.pos.shape[0])

    @classmethod
    def from_records(cls, records: Sequence[SiteRecord], samples: Sequence[str]) -> "SitesTable":
        contigs: List[str] = []
        for r in records:
            if not contigs or contigs[-1] != r.chrom:
                if r.chrom in contigs:
                    raise ValueError("sites records are not grouped by contig: %s" % r.chrom)
                contigs.append(r.chrom)
        t = cls(samples, contigs)
        n = len(records)
        ns = len(t.samples)
        t.pos = np.fromiter((r.start for r in records), dtype=np.int32, count=n)
        t.end = np.fromiter((r.end for r in records), dtype=np.int32, count=n)
        t.sflags = np.fromiter(
            (SFLAG_COMPLEX if r.is_complex else 0 for r in records), dtype=np.uint8, count=n
        )
        t.ref_base = np.fromiter(
            (0 if r.is_complex else ord(r.ref) for r in records), dtype=np.uint8, count=n
        )
        t.alt_base = np.fromiter(
            (0 if r.is_complex else ord(r.alts[0]) for r in records), dtype=np.uint8, count=n
        )
        t.gt = np.zeros((ns, n), dtype=np.uint8)
        t.ref_depth = np.zeros((ns, n), dtype=np.int32)
        t.alt_depth = np.zeros((ns, n), dtype=np.int32)
        t.gq = np.zeros((ns, n), dtype=np.float64)
        for i, r in enumerate(records):
            t.gt[:, i] = r.gt_types
            t.ref_depth[:, i] = r.ref_depths
            t.alt_depth[:, i] = r.alt_depths
            t.gq[:, i] = r.gt_quals
        t.ref_str = [r.ref for r in records]
        t.alt_strs = [list(r.alts) for r in records]
        chrom_ids = np.fromiter((t.contig_index[r.chrom] for r in records), dtype=np.int64, count=n)
        t.contig_off = np.searchsorted(chrom_ids, np.arange(len(contigs) + 1)).astype(np.int64)
        for c in range(len(contigs)):
            lo, hi = t.contig_off[c], t.contig_off[c + 1]
            if hi - lo > 1 and np.any(np.diff(t.pos[lo:hi]) < 0):
                raise ValueError("sites records are not sorted by position on %s" % contigs[c])
        return t

    def family_columns(self, kid: str, dad: str, mom: str):
        """Device-facing columns of one trio.

        Returns ``(gt_packed u8[S], ref_depth u16[3][S], alt_depth u16[3][S],
        gq u16[3][S])`` with member order kid, dad, mom.  gt_packed =
        kid | dad<<2 | mom<<4.  GQ is stored as floor(GQ): ``GQ < min_gt_qual``
        (reference informative_site_finder.py:64) with the integer threshold of
        ``--min-gt-qual`` (reference __main__.py:146-151) has the same truth value
        for GQ and floor(GQ).  Missing (-1, or any negative) -> 0xFFFF.
        A depth above 32767 does not fit the 16-bit columns (read as signed halfwords on the
        device): such sites are listed apart with their 32-bit depths (`wide_depths`, what
        abi.family_view / the engines take as `wide`) -- the reference takes any depth
        (informative_site_finder.py:46-73).  The 16-bit columns hold 32767 there.
        """
        cols = [self.samples.index(s) for s in (kid, dad, mom)]
        gtp = (
            (self.gt[cols[0]] & 3) | ((self.gt[cols[1]] & 3) << 2) | ((self.gt[cols[2]] & 3) << 4)
        ).astype(np.uint8)

        def depth16(a):
            a = a[cols]
            if a.size and a.min() < -1:
                raise ValueError("negative allele depth other than the missing marker -1")
            if a.size and a.max() > (1 << 30):
                raise OverflowError("allele depth above 2^30")
            out = np.minimum(a.astype(np.int64), U16_MAX_VALUE)
            out[out < 0] = U16_MISSING
            return np.ascontiguousarray(out.astype(np.uint16))

        rdc, adc = self.ref_depth[cols], self.alt_depth[cols]
        wide_sites = np.nonzero((rdc > U16_MAX_VALUE).any(axis=0) | (adc > U16_MAX_VALUE).any(axis=0))[0].astype(np.int64) if self.n_sites else np.zeros(0, np.int64)
        self.wide_depths = None
        if wide_sites.size:
            self.wide_depths = (wide_sites, np.ascontiguousarray(rdc[:, wide_sites].astype(np.int32)), np.ascontiguousarray(adc[:, wide_sites].astype(np.int32)))

        g = np.floor(self.gq[cols])
        g = np.where(np.isnan(g), -1.0, g)
        g = np.clip(g, -1, U16_MAX_VALUE).astype(np.int64)
        g[g < 0] = U16_MISSING
        return (
            np.ascontiguousarray(gtp),
            depth16(self.ref_depth),
            depth16(self.alt_depth),
            np.ascontiguousarray(g.astype(np.uint16)),
        )

    # ---- host-side interval queries (tabix semantics) -------------------
    def query(self, chrom: str, beg1: int, end1: int) -> np.ndarray:
        """Indices of records overlapping the 1-based inclusive interval, file order."""
        if chrom not in self.contig_index:
            return np.zeros(0, dtype=np.int64)
        c = self.contig_index[chrom]
        lo, hi = int(self.contig_off[c]), int(self.contig_off[c + 1])
        # record covers 1-based [start+1, end]; overlap: start+1 <= end1 and end >= beg1.  A record that reaches beg1 starts within
        # the longest record of the contig before it: the scan is a few records, not the contig up to the position
        span = self._max_span(c)
        # (keys in the column's own type: a Python int would make numpy convert the whole column for every look-up)
        key = self.pos.dtype.type
        lim = np.iinfo(self.pos.dtype)
        k_lo = lo + int(np.searchsorted(self.pos[lo:hi], key(min(max(beg1 - span, lim.min), lim.max)), side="left"))
        k_hi = lo + int(np.searchsorted(self.pos[lo:hi], key(min(max(end1 - 1, lim.min), lim.max)), side="right"))
        idx = np.arange(k_lo, k_hi, dtype=np.int64)
        if idx.size == 0:
            return idx
        return idx[self.end[k_lo:k_hi] >= beg1]

    def _max_span(self, c: int) -> int:
        """longest record (end - start) of contig c, cached"""
        cache = self.__dict__.setdefault("_span_cache", {})
        if c not in cache:
            lo, hi = int(self.contig_off[c]), int(self.contig_off[c + 1])
            cache[c] = int((np.asarray(self.end[lo:hi], np.int64) - np.asarray(self.pos[lo:hi], np.int64)).max()) if hi > lo else 1
        return cache[c]


class ReadsTable:
    """SoA columns of the decoded alignment records of one BAM (one kid)."""

    def __init__(self, contigs: Sequence[str]):
        self.contigs = list(contigs)
        self.contig_index = {c: i for i, c in enumerate(self.contigs)}
        nc = len(self.contigs)
        self.contig_off = np.zeros(nc + 1, dtype=np.int64)
        self.max_span = np.zeros(nc, dtype=np.int32)
        z32 = np.zeros(0, dtype=np.int32)
        self.start = z32
        self.end = z32
        self.flag = np.zeros(0, dtype=np.uint16)
        self.mapq = np.zeros(0, dtype=np.uint8)
        self.aux = np.zeros(0, dtype=np.uint8)
        self.tlen = z32
        self.qname = np.zeros(0, dtype=np.uint32)
        self.mate = z32
        self.cigar_off = np.zeros(0, dtype=np.uint32)
        self.n_cigar = np.zeros(0, dtype=np.uint16)
        self.cigar = np.zeros(0, dtype=np.uint32)
        self.l_seq = np.zeros(0, dtype=np.uint16)
        self.sq_off16 = np.zeros(0, dtype=np.uint32)
        self.seq = np.zeros(0, dtype=np.uint8)
        self.qual = np.zeros(0, dtype=np.uint8)
        self.qnames: List[str] = []  # id -> name

    @property
    def n_segs(self) -> int:
        return int(self.start.shape[0])

    @classmethod
    def from_segments(cls, segs: Sequence[Segment], contigs: Sequence[str]) -> "ReadsTable":
        """Build the table from records in BAM file order (sorted by tid, pos;
        records with tid < 0 are dropped: they are unreachable through fetch)."""
        t = cls(contigs)
        keep = [s for s in segs if s.tid >= 0]
        n = len(keep)
        tids = np.fromiter((s.tid for s in keep), dtype=np.int64, count=n)
        poss = np.fromiter((s.pos for s in keep), dtype=np.int64, count=n)
        if n > 1:
            key = tids * (1 << 40) + poss
            if np.any(np.diff(key) < 0):
                raise ValueError("alignment records are not coordinate sorted")
        t.contig_off = np.searchsorted(tids, np.arange(len(contigs) + 1)).astype(np.int64)
        t.start = poss.astype(np.int32)
        t.end = np.fromiter((s.endpos for s in keep), dtype=np.int32, count=n)
        t.flag = np.fromiter((s.flag for s in keep), dtype=np.uint16, count=n)
        t.mapq = np.fromiter((s.mapq for s in keep), dtype=np.uint8, count=n)
        t.tlen = np.fromiter((s.tlen for s in keep), dtype=np.int32, count=n)
        ids: Dict[str, int] = {}
        qn = np.zeros(n, dtype=np.uint32)
        for i, s in enumerate(keep):
            j = ids.get(s.qname)
            if j is None:
                j = len(ids)
                ids[s.qname] = j
                t.qnames.append(s.qname)
            qn[i] = j
        t.qname = qn
        aux = np.zeros(n, dtype=np.uint8)
        n_cigar = np.zeros(n, dtype=np.uint16)
        cigar_off = np.zeros(n, dtype=np.uint32)
        l_seq = np.zeros(n, dtype=np.uint16)
        sq_off16 = np.zeros(n, dtype=np.uint32)
        cig: List[int] = []
        sq_total = 0
        for i, s in enumerate(keep):
            a = 0
            if s.mtid == s.tid:
                a |= AUX_MATE_SAME_TID
            if s.has_sa:
                a |= AUX_HAS_SA
            if not s.cigar or not s.seq or s.qual is None:
                a |= AUX_DECODE_BAD
            aux[i] = a
            if len(s.cigar) > 0xFFFF or len(s.seq) > 0xFFFF:
                raise OverflowError("record too long for the 16-bit length columns")
            n_cigar[i] = len(s.cigar)
            cigar_off[i] = len(cig)
            cig.extend((l << 4) | op for op, l in s.cigar)
            l_seq[i] = len(s.seq) if s.seq else 0
            sq_off16[i] = sq_total >> 4
            sq_total += _pad16(int(l_seq[i]))
        if (sq_total >> 4) > 0xFFFFFFFF:
            raise OverflowError("sequence bytes exceed the 64 GiB row-offset range")
        t.aux, t.n_cigar, t.cigar_off, t.l_seq, t.sq_off16 = aux, n_cigar, cigar_off, l_seq, sq_off16
        t.cigar = np.array(cig, dtype=np.uint32)
        t.seq = np.zeros(sq_total, dtype=np.uint8)
        t.qual = np.zeros(sq_total, dtype=np.uint8)
        for i, s in enumerate(keep):
            L = int(l_seq[i])
            if L == 0:
                continue
            o = int(sq_off16[i]) << 4
            t.seq[o : o + L] = np.frombuffer(s.seq.encode("ascii"), dtype=np.uint8)
            if s.qual is not None:
                t.qual[o : o + L] = np.asarray(s.qual, dtype=np.uint8)
        t.mate = _link_mates(keep, t.end)
        nc = len(contigs)
        t.max_span = np.zeros(nc, dtype=np.int32)
        for c in range(nc):
            lo, hi = t.contig_off[c], t.contig_off[c + 1]
            if hi > lo:
                t.max_span[c] = int((t.end[lo:hi] - t.start[lo:hi]).max())
        return t


def _link_mates(segs: Sequence[Segment], endpos: np.ndarray) -> np.ndarray:
    """pysam.AlignmentFile.mate for every record (see module docstring)."""
    n = len(segs)
    by_name: Dict[str, List[int]] = {}
    for i, s in enumerate(segs):
        by_name.setdefault(s.qname, []).append(i)
    mate = np.full(n, -1, dtype=np.int32)
    for i, s in enumerate(segs):
        if not (s.flag & FPAIRED) or (s.flag & FMUNMAP) or s.mtid < 0:
            continue
        want = (s.flag ^ (FREAD1 | FREAD2)) & (FREAD1 | FREAD2)
        for j in by_name[s.qname]:  # ascending file order
            m = segs[j]
            if m.tid != s.mtid:
                continue
            if not (m.pos < s.mpos + 1 and int(endpos[j]) > s.mpos):
                continue
            if m.flag & want:
                mate[i] = j
                break
    return mate
