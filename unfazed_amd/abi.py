"""ctypes mirror of include/uz_types.h plus builders that wrap the host tables
(unfazed_amd.model) into those views.  No compute here."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from .model import ReadsTable, SitesTable

# status / origin codes (include/uz_types.h)
ST_OK, ST_NO_CAND, ST_NO_OVERLAP, ST_REF_EXCEPTION, ST_SKIPPED, ST_CAPACITY = 0, 1, 2, 3, 4, 5
OR_NONE, OR_DAD, OR_MOM, OR_AMBIGUOUS = 0, 1, 2, 3
VT_POINT, VT_DEL, VT_DUP, VT_OTHER_SV = 0, 1, 2, 3
DF_FETCH_FALLBACK = 1
CF_ALT_DAD = 1
CF_KA_SHIFT = 1
CL_HET, CL_CAND, CL_ALT_DAD = 1, 2, 4
CL_DEL_SHIFT, CL_DUP_SHIFT = 3, 5
ET_READBACKED, ET_ALLELE_BALANCE, ET_AMBIGUOUS_READBACKED, ET_AMBIGUOUS_ALLELE_BALANCE, ET_AMBIGUOUS_BOTH, ET_AMBIG_FLAG = 1, 2, 4, 8, 16, 32
ET_NAMES = [(ET_READBACKED, "READBACKED"), (ET_AMBIGUOUS_READBACKED, "AMBIGUOUS_READBACKED"), (ET_ALLELE_BALANCE, "ALLELE-BALANCE"),
            (ET_AMBIGUOUS_ALLELE_BALANCE, "AMBIGUOUS_ALLELE-BALANCE"), (ET_AMBIGUOUS_BOTH, "AMBIGUOUS_BOTH")]  # list order of summarize_record
FIND_WHOLE_REGION = 1
FIND_SECOND_WINDOW = 2

_p = C.c_void_p


class Params(C.Structure):
    _fields_ = [
        ("search_dist", C.c_int32),
        ("min_gt_qual", C.c_int32),
        ("min_depth", C.c_int32),
        ("min_map_qual", C.c_int32),
        ("readlen", C.c_int32),
        ("split_error_margin", C.c_int32),
        ("no_extended", C.c_int32),
        ("read_goal", C.c_int32),
        ("evidence_min_ratio", C.c_int32),
        ("reserved0", C.c_int32),
        ("ab_homref", C.c_double * 2),
        ("ab_homalt", C.c_double * 2),
        ("ab_het", C.c_double * 2),
    ]


def make_params(
    search_dist=5000,
    min_gt_qual=20,
    min_depth=10,
    min_map_qual=1,
    readlen=151,
    split_error_margin=5,
    no_extended=False,
    insert_size_max_sample=1000000,
    evidence_min_ratio=10,
    ab_homref=(0.0, 0.2),
    ab_homalt=(0.8, 1.0),
    ab_het=(0.2, 0.8),
) -> Params:
    """Defaults are the reference's CLI defaults (reference __main__.py:23-223)."""
    p = Params()
    p.search_dist = int(search_dist)
    p.min_gt_qual = int(min_gt_qual)
    p.min_depth = int(min_depth)
    p.min_map_qual = int(min_map_qual)
    p.readlen = int(readlen)
    p.split_error_margin = int(split_error_margin)
    p.no_extended = 1 if no_extended else 0
    p.read_goal = int(insert_size_max_sample)
    p.evidence_min_ratio = int(evidence_min_ratio)
    p.ab_homref[0], p.ab_homref[1] = float(ab_homref[0]), float(ab_homref[1])
    p.ab_homalt[0], p.ab_homalt[1] = float(ab_homalt[0]), float(ab_homalt[1])
    p.ab_het[0], p.ab_het[1] = float(ab_het[0]), float(ab_het[1])
    return p


class SitesView(C.Structure):
    _fields_ = [
        ("n_sites", C.c_int64),
        ("n_contigs", C.c_int32),
        ("reserved0", C.c_int32),
        ("contig_off", _p),
        ("pos", _p),
        ("sflags", _p),
        ("ref_base", _p),
        ("alt_base", _p),
    ]


class FamilyView(C.Structure):
    _fields_ = [
        ("gt", _p),
        ("ref_depth", _p * 3),
        ("alt_depth", _p * 3),
        ("gq", _p * 3),
        ("n_wide", C.c_int64),
        ("wide_site", _p),
        ("wide_ref_depth", _p * 3),
        ("wide_alt_depth", _p * 3),
        ("ref_depth8", _p * 3),  # the nine columns in eight bits for the host link (then ref_depth / alt_depth / gq are NULL)
        ("alt_depth8", _p * 3),
        ("gq8", _p * 3),
    ]


class ReadsView(C.Structure):
    _fields_ = [
        ("n_segs", C.c_int64),
        ("n_contigs", C.c_int32),
        ("reserved0", C.c_int32),
        ("contig_off", _p),
        ("max_span", _p),
        ("start", _p),
        ("end", _p),
        ("flag", _p),
        ("mapq", _p),
        ("aux", _p),
        ("tlen", _p),
        ("qname", _p),
        ("mate", _p),
        ("cigar_off", _p),
        ("n_cigar", _p),
        ("cigar", _p),
        ("l_seq", _p),
        ("sq_off16", _p),
        ("seq", _p),
        ("qual", _p),
        ("n_cigar_total", C.c_int64),
        ("n_sq_bytes", C.c_int64),
        ("n_qnames", C.c_uint32),
        ("reserved1", C.c_uint32),
    ]


class ReadsPackedView(C.Structure):
    """uz_reads_packed_view: the staged form of the alignment records (include/uz_types.h)"""
    _fields_ = [
        ("n_segs", C.c_int64),
        ("n_contigs", C.c_int32),
        ("min_base_qual", C.c_int32),
        ("contig_off", _p),
        ("max_span", _p),
        ("start", _p),
        ("end", _p),
        ("tlen", _p),
        ("mate", _p),
        ("qname", _p),
        ("flag", _p),
        ("l_seq", _p),
        ("n_cigar", _p),
        ("mapq", _p),
        ("aux", _p),
        ("cigar", _p),
        ("seq4", _p),
        ("qlow", _p),
        ("n_cigar_total", C.c_int64),
        ("n_row_units", C.c_int64),
        ("n_seq_units", C.c_int64),
        ("n_qnames", C.c_uint32),
        ("reserved1", C.c_uint32),
        # two-bit base rows for the host link (instead of seq4) + the listed bases that are not A/C/G/T
        ("seq2", _p),
        ("exc_rec", _p),
        ("exc_pos", _p),
        ("exc_code", _p),
        ("n_exc", C.c_int64),
        # the quality plane as lists for the host link (instead of qlow): count per record, positions for the records that
        # carry bases and have at most QLOW_LIST_MAX low bases
        ("n_low", _p),
        ("qlow_pos", _p),
        ("n_qlow_pos", C.c_int64),
        ("qlow_pos_wide", C.c_int32),
        ("cigar_compact", C.c_int32),  # 1: simple records (one M / = / X over the read) own no CIGAR word, their aux byte names the operation
        ("umask", _p),  # staged 32-base units per record (NULL: all)
        ("n_cigar_omitted", C.c_int64),
        # the small columns (flag, l_seq, n_cigar, mapq, aux, n_low) as a 16-bit index into a table of their combinations
        ("tup", _p), ("tup_flag", _p), ("tup_l_seq", _p), ("tup_n_cigar", _p), ("tup_mapq", _p), ("tup_aux", _p), ("tup_n_low", _p),
        ("n_tup", C.c_int64),
        ("tup_umask", _p),
        # start / tlen / mate / qname as 16-bit differences + an escape list
        ("start_d", _p), ("tlen_s", _p), ("mate_d", _p), ("qname_d", _p), ("esc16_key", _p), ("esc16_val", _p), ("n_esc16", C.c_int64),
        ("start_d8", _p),  # the start differences in eight bits (start_d then NULL)
        ("mate_d8", _p), ("qname_d8", _p),  # mate / name-id differences in eight bits (mate_d / qname_d then NULL)
        ("pair_d8", _p),  # the pair form: tlen, mate and name id in one byte (with start_d8; the other difference columns then NULL)
        # the bases of a record as a list (uz_types.h bl_*): per record the number of listed bases (0: its units travel as rows), their query
        # indices and two-bit codes
        ("bl_n", _p), ("tup_n_bl", _p), ("bl_pos", _p), ("bl_code", _p), ("n_bl", C.c_int64), ("n_bl_units", C.c_int64), ("bl_wide", C.c_int32),
        ("reserved2", C.c_int32),
        # span sums (uz_types.h pk_sums): the running sums the device's header build lays the records out by, from the packer
        ("pk_sums", _p), ("n_pk_spans", C.c_int64),
        # the dictionary index in one byte (uz_types.h tup8): the most frequent 255 combinations + an escape list, instead of `tup`
        ("tup8", _p), ("tup_hot", _p), ("tup_esc", _p), ("tup_esc_off", _p), ("n_tup_esc", C.c_int64),
    ]


PK_SUMS = 11


def pk_spans(n: int) -> int:
    """spans of the header build's passes over n records (uz_types.h: UZ_PK_SHIFT)"""
    shift = 12 if (int(n) >> 12) >= 4096 else 10
    return (int(n) + (1 << shift) - 1) >> shift


AUX_NO_SEQ = 8
AUX_SIMPLE_MASK = 48  # cigar_compact: the record's one M / = / X operation is named by its aux byte
# per-record columns of the packed view and their element types
PACKED_RECORD_COLS = [("start", np.int32), ("end", np.int32), ("tlen", np.int32), ("mate", np.int32), ("qname", np.uint32),
                      ("flag", np.uint16), ("l_seq", np.uint16), ("n_cigar", np.uint16), ("mapq", np.uint8), ("aux", np.uint8)]
SEQ4_UNIT_BYTES, QLOW_UNIT_BYTES, SEQ2_UNIT_BYTES = 16, 4, 8
QLOW_LIST_MAX = 10
UMASK_ALL = 0xFFFF


def row_units(l_seq):
    return (np.asarray(l_seq).astype(np.int64) + 31) >> 5


TUP8_SPAN = 1024  # uz_types.h UZ_TUP8_SPAN


def packed_view_alloc(n, n_contigs, n_cigar_total, n_row_units, alloc=None, n_seq_units=None, n_exc=None, n_qlow_pos=None,
                      qlow_pos_wide=False, with_end=True, with_umask=False, cigar_omitted=None, n_tup=None, n_esc16=None, start8=False,
                      narrow8=False, pair8=False, n_bl=None, n_bl_units=0, bl_wide=False, pk_sums=False, tup_off_link=False) -> "Held":
    """A packed view over freshly allocated, writable arrays (alloc(nbytes) -> uint8 array; default numpy).
    n_seq_units: row units of the records that carry bases (default: all of them).
    n_exc: None = four-bit base rows (seq4); a number = two-bit rows (seq2) with that many listed bases (exc_*).
    n_qlow_pos: None = the quality plane (qlow); a number = per-record counts (n_low) + that many listed positions (qlow_pos).
    with_end: False leaves the `end` column out (the device derives it from the CIGAR, as a BAM decoder did).
    with_umask: a per-record mask of the staged 32-base units (n_seq_units then counts staged units).
    n_tup: None = the plain small columns; a number = the dictionary form with that many combinations (tup + tup_* instead of flag,
    l_seq, n_cigar, mapq, aux and n_low).
    n_esc16: None = start / tlen / mate / qname as 32-bit columns; a number = as 16-bit differences with that many escapes.
    pk_sums: room for the span sums (uz_types.h pk_sums; io_native.block_sums fills them).
    pair8 (with start8): tlen, mate and name id in the pair form's one byte (pair_d8) instead of tlen_s / mate_d* / qname_d*.
    n_bl: None = every record's staged units travel as rows; a number = the list form of the bases for some records (bl_* columns with that many
    listed bases; the count per record in the dictionary, tup_n_bl, or the plain bl_n column), n_bl_units their row units.
    cigar_omitted: None = every CIGAR word; a number = cigar_compact with that many simple records (n_cigar_total is the plain total:
    the words that stay home are taken off here)."""
    if n_seq_units is None:
        n_seq_units = n_row_units
    if alloc is None:
        alloc = lambda nbytes: np.zeros(max(16, nbytes), dtype=np.uint8)  # noqa: E731
    arrs = {}
    TUP_COLS = ("flag", "l_seq", "n_cigar", "mapq", "aux")
    for name, dt in PACKED_RECORD_COLS:
        if name == "end" and not with_end:
            continue
        if n_tup is not None and name in TUP_COLS:
            continue
        if n_esc16 is not None and name in ("start", "tlen", "mate", "qname"):
            continue
        arrs[name] = alloc(max(1, n) * np.dtype(dt).itemsize)[: max(1, n) * np.dtype(dt).itemsize].view(dt)
    if with_umask and n_tup is None:
        arrs["umask"] = alloc(2 * max(1, n))[: 2 * max(1, n)].view(np.uint16)
    arrs["contig_off"] = alloc(8 * (n_contigs + 1))[: 8 * (n_contigs + 1)].view(np.int64)
    arrs["max_span"] = alloc(4 * max(1, n_contigs))[: 4 * max(1, n_contigs)].view(np.int32)
    if cigar_omitted is not None:
        n_cigar_total = n_cigar_total - cigar_omitted
    arrs["cigar"] = alloc(4 * max(1, n_cigar_total))[: 4 * max(1, n_cigar_total)].view(np.uint32)
    if n_exc is None:
        arrs["seq4"] = alloc(SEQ4_UNIT_BYTES * max(1, n_seq_units))[: SEQ4_UNIT_BYTES * max(1, n_seq_units)]
    else:
        arrs["seq2"] = alloc(SEQ2_UNIT_BYTES * max(1, n_seq_units))[: SEQ2_UNIT_BYTES * max(1, n_seq_units)]
        arrs["exc_rec"] = alloc(4 * max(1, n_exc))[: 4 * max(1, n_exc)].view(np.uint32)
        arrs["exc_pos"] = alloc(2 * max(1, n_exc))[: 2 * max(1, n_exc)].view(np.uint16)
        arrs["exc_code"] = alloc(max(1, n_exc))[: max(1, n_exc)]
    if n_qlow_pos is None:
        arrs["qlow"] = alloc(QLOW_UNIT_BYTES * max(1, n_row_units))[: QLOW_UNIT_BYTES * max(1, n_row_units)]
    else:
        if n_tup is None:
            arrs["n_low"] = alloc(max(1, n))[: max(1, n)]
        w = 2 if qlow_pos_wide else 1
        arrs["qlow_pos"] = alloc(w * max(1, n_qlow_pos))[: w * max(1, n_qlow_pos)]
    v = ReadsPackedView()
    v.n_segs, v.n_contigs, v.n_cigar_total, v.n_row_units, v.n_seq_units = n, n_contigs, n_cigar_total, n_row_units, n_seq_units
    v.n_exc = 0 if n_exc is None else n_exc
    if n_tup is not None:
        # (tup_off_link: the 16-bit index is an intermediate -- compact_tup() replaces it by the one-byte form -- and stays out of the caller's
        # page-locked block)
        arrs["tup"] = (np.zeros(2 * max(1, n), np.uint8) if tup_off_link else alloc(2 * max(1, n)))[: 2 * max(1, n)].view(np.uint16)
        for name, dt in (("tup_flag", np.uint16), ("tup_l_seq", np.uint16), ("tup_n_cigar", np.uint16), ("tup_mapq", np.uint8),
                         ("tup_aux", np.uint8)) + ((("tup_n_low", np.uint8),) if n_qlow_pos is not None else ()) + (
                                 (("tup_umask", np.uint16),) if with_umask else ()):
            arrs[name] = alloc(np.dtype(dt).itemsize * max(1, n_tup))[: np.dtype(dt).itemsize * max(1, n_tup)].view(dt)
        v.n_tup = n_tup
    if n_bl is not None:
        if n_tup is not None:
            arrs["tup_n_bl"] = alloc(max(1, n_tup))[: max(1, n_tup)]
        else:
            arrs["bl_n"] = alloc(max(1, n))[: max(1, n)]
        wb = 2 if bl_wide else 1
        arrs["bl_pos"] = alloc(wb * max(1, n_bl))[: wb * max(1, n_bl)]
        arrs["bl_code"] = alloc((max(1, n_bl) + 3) // 4 + 4)[: (max(1, n_bl) + 3) // 4]
        v.n_bl, v.n_bl_units, v.bl_wide = n_bl, n_bl_units, 1 if bl_wide else 0
    if n_esc16 is not None:
        if start8:
            arrs["start_d8"] = alloc(max(1, n))[: max(1, n)]
        if pair8:
            assert start8 and not narrow8
            arrs["pair_d8"] = alloc(max(1, n))[: max(1, n)]
        if narrow8:  # (with start8) mate / name-id differences as signed bytes
            assert start8
            for name in ("mate_d8", "qname_d8"):
                arrs[name] = alloc(max(1, n))[: max(1, n)].view(np.int8)
        for name in (() if pair8 else ("tlen_s",) if narrow8 else ("tlen_s", "mate_d", "qname_d")) if start8 else ("start_d", "tlen_s", "mate_d", "qname_d"):
            arrs[name] = alloc(2 * max(1, n))[: 2 * max(1, n)].view(np.int16)
        arrs["esc16_key"] = alloc(8 * max(1, n_esc16))[: 8 * max(1, n_esc16)].view(np.uint64)
        arrs["esc16_val"] = alloc(4 * max(1, n_esc16))[: 4 * max(1, n_esc16)].view(np.int32)
        v.n_esc16 = n_esc16
    if pk_sums:  # the span sums ride along (filled by io_native.block_sums once the columns are)
        nb = pk_spans(n)
        arrs["pk_sums"] = alloc(8 * PK_SUMS * (nb + 1))[: 8 * PK_SUMS * (nb + 1)].view(np.uint64)
        v.n_pk_spans = nb
    v.cigar_compact = 0 if cigar_omitted is None else 1
    v.n_cigar_omitted = 0 if cigar_omitted is None else cigar_omitted
    v.n_qlow_pos = 0 if n_qlow_pos is None else n_qlow_pos
    v.qlow_pos_wide = 1 if qlow_pos_wide else 0
    for k, a in arrs.items():
        setattr(v, k, a.ctypes.data)
    return Held(v, arrs)


def compact_tup(held: "Held", alloc=None) -> bool:
    """The dictionary index of a packed view from two bytes per record to one (uz_types.h tup8): the 255 most frequent combinations by a byte,
    the rest through an escape list.  In place: `tup` leaves the view, tup8 / tup_hot / tup_esc / tup_esc_off (from `alloc`) enter it.
    -> False (view untouched) when it has no dictionary form."""
    a, v = held.arrays, held.view
    if "tup" not in a or not v.tup:
        return False
    if alloc is None:
        alloc = lambda nbytes: np.zeros(max(16, nbytes), dtype=np.uint8)  # noqa: E731
    n, n_tup = int(v.n_segs), int(v.n_tup)
    tup = a["tup"][:n]
    cnt = np.bincount(tup, minlength=max(1, n_tup))
    hot = np.argsort(-cnt, kind="stable")[:255]
    hot = hot[cnt[hot] > 0]
    lut = np.full(max(1, cnt.size), 255, np.uint8)
    lut[hot] = np.arange(hot.size, dtype=np.uint8)
    t8 = alloc(max(1, n))[: max(1, n)]
    t8[:n] = lut[tup]
    esc_mask = t8[:n] == 255
    n_esc = int(esc_mask.sum())
    esc = alloc(2 * max(1, n_esc))[: 2 * max(1, n_esc)].view(np.uint16)
    esc[:n_esc] = tup[esc_mask]
    nsp = (n + TUP8_SPAN - 1) // TUP8_SPAN
    off = alloc(4 * (nsp + 1))[: 4 * (nsp + 1)].view(np.uint32)
    off[0] = 0
    if nsp:
        off[1:] = np.cumsum(np.add.reduceat(esc_mask.astype(np.int64), np.arange(0, n, TUP8_SPAN)))
    hot16 = alloc(512)[:512].view(np.uint16)
    hot16[:] = 0
    hot16[: hot.size] = hot.astype(np.uint16)
    del a["tup"]
    a.update(tup8=t8, tup_hot=hot16, tup_esc=esc, tup_esc_off=off)
    v.tup = None
    v.tup8, v.tup_hot, v.tup_esc, v.tup_esc_off, v.n_tup_esc = t8.ctypes.data, hot16.ctypes.data, esc.ctypes.data, off.ctypes.data, n_esc
    return True


def wide_columns(held: "Held") -> dict:
    """start, tlen, mate, qname of a packed view as plain per-record arrays, whichever way it carries them."""
    a, n = held.arrays, int(held.view.n_segs)
    if "start_d" not in a and "start_d8" not in a:
        return {k: a[k][:n] for k in ("start", "tlen", "mate", "qname")}
    ne = int(held.view.n_esc16)
    key, val = a["esc16_key"][:ne], a["esc16_val"][:ne].astype(np.int64)
    assert np.all(np.diff(key.astype(np.int64)) > 0)
    out = {}
    if "pair_d8" in a:
        return _pair_columns(held, key, val)
    for col, (name, src) in enumerate((("start", "start_d8" if "start_d8" in a else "start_d"), ("tlen", "tlen_s"),
                                       ("mate", "mate_d8" if "mate_d8" in a else "mate_d"), ("qname", "qname_d8" if "qname_d8" in a else "qname_d"))):
        v = a[src][:n].astype(np.int64)
        sel = (key & np.uint64(3)) == col
        rec = (key[sel] >> np.uint64(2)).astype(np.int64)
        assert np.array_equal(np.nonzero(v == (255 if src == "start_d8" else -128 if src.endswith("_d8") else -32768))[0], rec)
        v[rec] = val[sel]
        if name in ("start", "qname"):
            v = np.cumsum(v) & 0xFFFFFFFF
        elif name == "mate":
            none = a[src][:n] == (-127 if src == "mate_d8" else -32767)
            esc = np.zeros(n, bool)
            esc[rec] = True
            v = np.where(none, -1, np.where(esc, v, v + np.arange(n)))
        out[name] = v.astype(np.uint32 if name == "qname" else np.int32)
    return out


def record_ends(held: "Held", start: np.ndarray) -> np.ndarray:
    """`end` of every record of a packed view: the column, or what the device derives when it was left out (htslib's bam_endpos)"""
    a, n = held.arrays, int(held.view.n_segs)
    if "end" in a:
        return a["end"][:n].astype(np.int64)
    sm = small_columns(held)
    flag, ncig, lseq, aux = (sm[k].astype(np.int64) for k in ("flag", "n_cigar", "l_seq", "aux"))
    compact = bool(held.view.cigar_compact)
    simple = ((aux & AUX_SIMPLE_MASK) != 0) if compact else np.zeros(n, bool)
    words_of = np.where(simple, 0, ncig)
    off = np.concatenate([[0], np.cumsum(words_of)])
    cig = a["cigar"]
    ref_len = np.where(simple, lseq, 0).astype(np.int64)
    for i in np.nonzero(~simple & (ncig > 0))[0]:
        w = cig[off[i]: off[i] + ncig[i]].astype(np.int64)
        op = w & 15
        ref_len[i] = int((w >> 4)[(op == 0) | (op == 2) | (op == 3) | (op == 7) | (op == 8)].sum())
    unmapped = (flag & 4) != 0
    return np.where(unmapped | (ncig == 0), start + 1, start + np.maximum(ref_len, 1))


def _pair_columns(held: "Held", key, val) -> dict:
    """the pair form (uz_types.h, pair_d8) back to start / tlen / mate / qname, as the device rebuilds them"""
    a, n = held.arrays, int(held.view.n_segs)
    def esc(col):
        sel = (key & np.uint64(3)) == col
        return (key[sel] >> np.uint64(2)).astype(np.int64), val[sel]
    sd = a["start_d8"][:n].astype(np.int64)
    rec, v = esc(0)
    assert np.array_equal(np.nonzero(sd == 255)[0], rec)
    sd[rec] = v
    start = (np.cumsum(sd) & 0xFFFFFFFF).astype(np.uint32).view(np.int32).astype(np.int64)
    p = a["pair_d8"][:n].astype(np.int64)
    first, second, given, new, old = (p >= 1) & (p <= 252), (p == 0) | (p == 253), p == 253, p == 254, p == 255
    other = new | old
    assert np.array_equal(np.nonzero(other | given)[0], esc(1)[0]) and np.array_equal(np.nonzero(other)[0], esc(2)[0])
    rec3, v3 = esc(3)
    assert np.array_equal(np.nonzero(old)[0], rec3)
    qname = np.cumsum(first | new) - 1
    qname[rec3] = v3
    mate = np.full(n, -1, np.int64)
    tlen = np.zeros(n, np.int64)
    tlen[other | given] = esc(1)[1]
    mate[other] = esc(2)[1]
    fi = np.nonzero(first)[0]
    fj = fi + p[fi]
    assert fi.size == int(second.sum()) and (fj.size == 0 or fj.max() < n) and np.all(second[fj]) and np.unique(fj).size == fj.size
    end = record_ends(held, start)
    span = np.where(given[fj], -tlen[fj], np.maximum(end[fi], end[fj]) - start[fi])
    mate[fi], mate[fj] = fj, fi
    tlen[fi], tlen[fj] = span, -span
    qname[fj] = qname[fi]
    return {"start": start.astype(np.int32), "tlen": tlen.astype(np.int32), "mate": mate.astype(np.int32), "qname": qname.astype(np.uint32)}


def tup_column(held: "Held") -> np.ndarray:
    """the 16-bit dictionary index of every record, from either form (tup, or tup8 + hot table + escape list: what the device rebuilds)"""
    a, n = held.arrays, int(held.view.n_segs)
    if "tup" in a:
        return a["tup"][:n]
    t8 = a["tup8"][:n]
    out = a["tup_hot"][t8.astype(np.int64)].astype(np.uint16)
    esc = t8 == 255
    assert int(esc.sum()) == int(held.view.n_tup_esc)
    out[esc] = a["tup_esc"][: int(held.view.n_tup_esc)]
    return out


def small_columns(held: "Held") -> dict:
    """flag, l_seq, n_cigar, mapq, aux (and n_low) of a packed view as plain per-record arrays, whichever way it carries them
    (the dictionary form keeps a 16-bit index per record and a table of the combinations)."""
    a, n = held.arrays, int(held.view.n_segs)
    names = ["flag", "l_seq", "n_cigar", "mapq", "aux"] + (["n_low"] if ("n_low" in a or "tup_n_low" in a) else []) + (
        ["umask"] if ("umask" in a or "tup_umask" in a) else [])
    if "tup" not in a and "tup8" not in a:
        out = {k: a[k][:n] for k in names}
        if "bl_n" in a:
            out["bl_n"] = a["bl_n"][:n]
        return out
    t = tup_column(held).astype(np.int64)
    assert n == 0 or t.max() < int(held.view.n_tup)
    out = {k: a["tup_" + k][t] for k in names}
    if "tup_n_bl" in a:  # the list form of the bases: listed bases per record (0: its units travel as rows)
        out["bl_n"] = a["tup_n_bl"][t]
    return out


def base_lists(held: "Held"):
    """the list form of the bases of a packed view as (offsets [n + 1], query indices, two-bit codes), or None when it has none"""
    a, n = held.arrays, int(held.view.n_segs)
    if "bl_pos" not in a:
        return None
    cnt = small_columns(held)["bl_n"].astype(np.int64)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    m = int(off[-1])
    assert m == int(held.view.n_bl)
    pos = a["bl_pos"][: 2 * m].view(np.uint16)[:m] if held.view.bl_wide else a["bl_pos"][:m].astype(np.uint16)
    e = np.arange(m)
    code = (a["bl_code"][e >> 2] >> (2 * (e & 3))) & 3 if m else np.zeros(0, np.uint8)
    return off, pos.astype(np.uint16), code.astype(np.uint8)


class CohortGroup(C.Structure):
    """uz_cohort_group: one kid of a cohort batch"""
    _fields_ = [("fam_id", C.c_int32), ("reads_id", C.c_int32), ("dnm_first", C.c_int32), ("dnm_count", C.c_int32),
                ("cutoff", C.c_double)]


class DnmsView(C.Structure):
    _fields_ = [
        ("n", C.c_int32),
        ("reserved0", C.c_int32),
        ("contig", _p),
        ("rcontig", _p),
        ("start", _p),
        ("end", _p),
        ("vartype", _p),
        ("dflags", _p),
        ("mult", _p),
        ("allele_off", _p),
        ("alleles", _p),
        ("cutoff", C.c_double),
    ]


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data if a.size else 0


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


class Held:
    """A ctypes view plus the numpy arrays it points into (kept alive)."""

    def __init__(self, view, arrays):
        self.view = view
        self.arrays = arrays

    def ref(self):
        return C.byref(self.view)


def sites_view(t: SitesTable) -> Held:
    arrs = dict(
        contig_off=_c(t.contig_off, np.int64),
        pos=_c(t.pos, np.int32),
        sflags=_c(t.sflags, np.uint8),
        ref_base=_c(t.ref_base, np.uint8),
        alt_base=_c(t.alt_base, np.uint8),
    )
    v = SitesView()
    v.n_sites = t.n_sites
    v.n_contigs = len(t.contigs)
    for k, a in arrs.items():
        setattr(v, k, _ptr(a))
    return Held(v, arrs)


def family_view(gt: np.ndarray, rd: np.ndarray, ad: np.ndarray, gq: np.ndarray, wide=None) -> Held:
    """gt u8[S]; rd/ad/gq u16[3][S] in kid, dad, mom order.  wide = (site int64[W], ref_depth int32[3][W], alt_depth int32[3][W]): the
    sites whose depths do not fit the 16-bit columns (model.SitesTable.family_columns lists them), or None."""
    gt = _c(gt, np.uint8)
    eight = all(np.asarray(x[m]).dtype == np.uint8 for x in (rd, ad, gq) for m in range(3))  # the link form of family_columns8
    dt = np.uint8 if eight else np.uint16
    rd = [_c(rd[m], dt) for m in range(3)]
    ad = [_c(ad[m], dt) for m in range(3)]
    gq = [_c(gq[m], dt) for m in range(3)]
    v = FamilyView()
    v.gt = _ptr(gt)
    for m in range(3):
        if eight:
            v.ref_depth8[m], v.alt_depth8[m], v.gq8[m] = _ptr(rd[m]), _ptr(ad[m]), _ptr(gq[m])
        else:
            v.ref_depth[m], v.alt_depth[m], v.gq[m] = _ptr(rd[m]), _ptr(ad[m]), _ptr(gq[m])
    keep = dict(gt=gt, rd=rd, ad=ad, gq=gq)
    if wide is not None and len(wide[0]):
        ws = _c(wide[0], np.int64)
        wr = [_c(wide[1][m], np.int32) for m in range(3)]
        wa = [_c(wide[2][m], np.int32) for m in range(3)]
        v.n_wide, v.wide_site = int(ws.size), _ptr(ws)
        for m in range(3):
            v.wide_ref_depth[m], v.wide_alt_depth[m] = _ptr(wr[m]), _ptr(wa[m])
        keep.update(wide_site=ws, wide_rd=wr, wide_ad=wa)
    return Held(v, keep)


def family_columns8(rd, ad, gq, wide=None):
    """The 16-bit family columns (rd / ad / gq u16[3][S], 0xFFFF = missing; wide as family_view takes it) in the eight-bit link form of
    uz_family_view: -> (rd8, ad8, gq8 as uint8 [3][S], wide) -- depth bytes: 0 .. 253 the depth, 254 missing, 255 = the site is in the wide
    list with its exact depths (any member's depth of 254 or more puts it there); quality bytes: min(gq, 254), 255 missing."""
    rd = np.asarray(rd, np.uint16)
    ad = np.asarray(ad, np.uint16)
    gq = np.asarray(gq, np.uint16)
    S = rd.shape[1]
    big = (((rd >= 254) & (rd != 0xFFFF)) | ((ad >= 254) & (ad != 0xFFFF))).any(axis=0)
    listed = np.zeros(S, bool)
    w_site = np.zeros(0, np.int64)
    if wide is not None and len(wide[0]):
        w_site = np.asarray(wide[0], np.int64)
        listed[w_site] = True
    new = np.nonzero(big & ~listed)[0]
    exact = lambda col: np.where(col == 0xFFFF, -1, col.astype(np.int64)).astype(np.int32)  # noqa: E731
    if new.size or w_site.size:
        sites = np.concatenate([w_site, new])
        order = np.argsort(sites, kind="stable")
        wr = [np.concatenate([np.asarray(wide[1][m], np.int32) if w_site.size else np.zeros(0, np.int32), exact(rd[m][new])])[order] for m in range(3)]
        wa = [np.concatenate([np.asarray(wide[2][m], np.int32) if w_site.size else np.zeros(0, np.int32), exact(ad[m][new])])[order] for m in range(3)]
        wide = (sites[order], wr, wa)
        listed[new] = True
    rd8 = np.where(listed[None, :], 255, np.where(rd == 0xFFFF, 254, rd)).astype(np.uint8)
    ad8 = np.where(listed[None, :], 255, np.where(ad == 0xFFFF, 254, ad)).astype(np.uint8)
    gq8 = np.where(gq == 0xFFFF, 255, np.minimum(gq, 254)).astype(np.uint8)
    return rd8, ad8, gq8, wide


def reads_view(t: ReadsTable) -> Held:
    arrs = dict(
        contig_off=_c(t.contig_off, np.int64),
        max_span=_c(t.max_span, np.int32),
        start=_c(t.start, np.int32),
        end=_c(t.end, np.int32),
        flag=_c(t.flag, np.uint16),
        mapq=_c(t.mapq, np.uint8),
        aux=_c(t.aux, np.uint8),
        tlen=_c(t.tlen, np.int32),
        qname=_c(t.qname, np.uint32),
        mate=_c(t.mate, np.int32),
        cigar_off=_c(t.cigar_off, np.uint32),
        n_cigar=_c(t.n_cigar, np.uint16),
        cigar=_c(t.cigar, np.uint32),
        l_seq=_c(t.l_seq, np.uint16),
        sq_off16=_c(t.sq_off16, np.uint32),
        seq=_c(t.seq, np.uint8),
        qual=_c(t.qual, np.uint8),
    )
    v = ReadsView()
    v.n_segs = t.n_segs
    v.n_contigs = len(t.contigs)
    for k, a in arrs.items():
        setattr(v, k, _ptr(a))
    v.n_cigar_total = int(arrs["cigar"].shape[0])
    v.n_sq_bytes = int(arrs["seq"].shape[0])
    v.n_qnames = int(t.qname.max()) + 1 if t.n_segs else 0
    return Held(v, arrs)


def dnms_view(
    contig: Sequence[int],
    rcontig: Sequence[int],
    start: Sequence[int],
    end: Sequence[int],
    vartype: Sequence[int],
    refs: Sequence[bytes],
    alts: Sequence[bytes],
    cutoff: float,
    dflags: Optional[Sequence[int]] = None,
    mult: Optional[Sequence[int]] = None,
) -> Held:
    n = len(start)
    # REF then ALT of every DNM, back to back: the offsets from the lengths in one pass (a store per DNM into a numpy array was 20 ms per 20 k DNMs)
    lens = np.empty(2 * n, dtype=np.int64)
    lens[0::2] = np.fromiter(map(len, refs), np.int64, n)
    lens[1::2] = np.fromiter(map(len, alts), np.int64, n)
    off = np.zeros(2 * n + 1, dtype=np.uint32)
    np.cumsum(lens, out=lens)
    assert n == 0 or lens[-1] < (1 << 32)
    off[1:] = lens
    alleles = np.frombuffer(b"".join(x for pair in zip(refs, alts) for x in pair) + b"\0", dtype=np.uint8).copy()
    arrs = dict(
        contig=_c(contig, np.int32),
        rcontig=_c(rcontig, np.int32),
        start=_c(start, np.int32),
        end=_c(end, np.int32),
        vartype=_c(vartype, np.uint8),
        dflags=_c(dflags if dflags is not None else np.zeros(n), np.uint8),
        mult=_c(mult if mult is not None else np.ones(n), np.uint8),
        allele_off=off,
        alleles=alleles,
    )
    v = DnmsView()
    v.n = n
    for k, a in arrs.items():
        setattr(v, k, _ptr(a))
    v.cutoff = float(cutoff)
    return Held(v, arrs)
