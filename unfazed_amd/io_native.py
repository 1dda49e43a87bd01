"""ctypes binding of libunfazed_io.so (include/unfazed_io.h): the native, multi-threaded BAM and
sites-VCF decoders.  They produce the same column tables as the Python decoders in io_bam.py /
io_vcf.py (kept as the readable statement of the formats and as the checker in
tests/test_io_native.py); the columns are numpy views of the memory the handle owns.

SURVEY.md section 8(f)-2.  Reference seams: pysam.AlignmentFile / .mate() in
unfazed/read_collector.py:11-25, :372-392, :402 and cyvcf2.VCF in
unfazed/informative_site_finder.py:213, :571.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import List, Sequence

import numpy as np

from . import abi
from .model import ReadsTable, SitesTable

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

IO_EXPORTS = [
    "uz_io_last_error", "uz_bam_decode", "uz_bam_free", "uz_bam_n_contigs", "uz_bam_contig_name",
    "uz_bam_contig_length", "uz_bam_n_file_records", "uz_bam_n_records", "uz_bam_view", "uz_bam_qname",
    "uz_bam_tlen_head", "uz_bam_timing", "uz_bam_decode_regions", "uz_bam_io_stats", "uz_vcf_decode", "uz_vcf_free", "uz_vcf_view_get", "uz_vcf_sample",
    "uz_vcf_contig", "uz_vcf_ref", "uz_vcf_alt", "uz_vcf_header", "uz_vcf_line", "uz_vcf_info", "uz_vcf_is_bcf",
    "uz_reads_pack_sizes", "uz_reads_pack_exceptions", "uz_reads_pack_lists", "uz_reads_pack_end_derivable", "uz_reads_pack_cigar_omitted", "uz_reads_pack", "uz_reads_source_open", "uz_reads_source_close", "uz_reads_select_plan",
    "uz_select_n_records", "uz_select_n_cigar_total", "uz_select_n_row_units", "uz_select_n_seq_units", "uz_select_n_bl", "uz_select_n_bl_units", "uz_select_bl_wide", "uz_select_n_exc", "uz_select_n_qlow_pos", "uz_select_qlow_pos_wide", "uz_select_end_derivable", "uz_select_n_cigar_omitted", "uz_select_n_tuples", "uz_select_n_esc16", "uz_select_n_esc16_start8", "uz_select_n_esc16_narrow8", "uz_select_pair8_ok", "uz_select_n_esc16_pair8", "uz_select_n_new_names", "uz_select_qname_map",
    "uz_reads_select_fill", "uz_select_free", "uz_vcf_decode_regions", "uz_vcf_index_names", "uz_vcf_io_stats",
    "uz_bam_decode_memory",
    "uz_bamsrc_open", "uz_bamsrc_close", "uz_bamsrc_n_contigs", "uz_bamsrc_contig_name", "uz_bamsrc_contig_length", "uz_bamsrc_tlen_head",
    "uz_index_summary", "uz_inflate_backend", "uz_io_default_threads", "uz_io_cpu_quota", "uz_bam_stage_plan", "uz_bam_stage_begin", "uz_bam_stage_finish", "uz_stage_gather_blocks", "uz_stage_set_inflated", "uz_stage_sizes", "uz_stage_io_stats", "uz_stage_timing", "uz_stage_fill", "uz_stage_qname", "uz_stage_qnames",
    "uz_stage_free", "uz_stage_walk_plan_sizes", "uz_stage_walk_plan", "uz_bam_stage_finish_desc", "uz_stage_kept_sizes", "uz_stage_kept", "uz_stage_walk_host", "uz_stage_kept_debug", "uz_stage_name_records", "uz_packed_block_sums", "uz_stage_merge_subtasks", "uz_bam_stage_finish_sub",
    "uz_stage_walk_flagged", "uz_stage_lookup", "uz_stage_extra", "uz_stage_n_lookup_tasks",
]


class VcfView(C.Structure):
    _fields_ = [
        ("n_sites", C.c_int64), ("n_samples", C.c_int32), ("n_contigs", C.c_int32),
        ("contig_off", C.c_void_p), ("pos", C.c_void_p), ("end", C.c_void_p), ("sflags", C.c_void_p),
        ("ref_base", C.c_void_p), ("alt_base", C.c_void_p), ("gt", C.c_void_p), ("ref_depth", C.c_void_p),
        ("alt_depth", C.c_void_p), ("gq", C.c_void_p),
    ]


def lib_path() -> str:
    return os.environ.get("UZ_IO_LIB", os.path.join(_HERE, "libunfazed_io.so"))


def load():
    """The decoder library; built on first use when the tree has no copy yet (g++ and zlib only)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        from . import build
        build.build_io()
    lib = C.CDLL(path)
    lib.uz_io_last_error.restype = C.c_char_p
    lib.uz_bam_decode.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.uz_bam_free.argtypes = [C.c_void_p]
    lib.uz_bam_free.restype = None
    lib.uz_bam_decode_regions.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                          C.POINTER(C.c_void_p)]
    lib.uz_bam_io_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.uz_bam_io_stats.restype = None
    lib.uz_bam_n_contigs.argtypes = [C.c_void_p]
    lib.uz_bam_contig_name.argtypes = [C.c_void_p, C.c_int32]
    lib.uz_bam_contig_name.restype = C.c_char_p
    lib.uz_bam_contig_length.argtypes = [C.c_void_p, C.c_int32]
    lib.uz_bam_n_file_records.argtypes = [C.c_void_p]
    lib.uz_bam_n_file_records.restype = C.c_int64
    lib.uz_bam_n_records.argtypes = [C.c_void_p]
    lib.uz_bam_n_records.restype = C.c_int64
    lib.uz_bam_view.argtypes = [C.c_void_p, C.POINTER(abi.ReadsView)]
    lib.uz_bam_qname.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
    lib.uz_bam_qname.restype = C.c_void_p
    lib.uz_bam_tlen_head.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.uz_bam_tlen_head.restype = C.c_int64
    lib.uz_bam_decode_memory.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    lib.uz_bam_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.uz_bam_timing.restype = None
    lib.uz_vcf_decode.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.uz_vcf_decode_regions.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.uz_vcf_index_names.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.uz_vcf_index_names.restype = C.c_int64
    lib.uz_vcf_io_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.uz_vcf_io_stats.restype = None
    lib.uz_vcf_free.argtypes = [C.c_void_p]
    lib.uz_vcf_free.restype = None
    lib.uz_vcf_view_get.argtypes = [C.c_void_p, C.POINTER(VcfView)]
    lib.uz_vcf_sample.argtypes = [C.c_void_p, C.c_int32]
    lib.uz_vcf_sample.restype = C.c_char_p
    lib.uz_vcf_contig.argtypes = [C.c_void_p, C.c_int32]
    lib.uz_vcf_contig.restype = C.c_char_p
    for fn in (lib.uz_vcf_ref, lib.uz_vcf_alt, lib.uz_vcf_line):
        fn.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]
        fn.restype = C.c_void_p
    lib.uz_vcf_info.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(C.c_int32)]
    lib.uz_vcf_info.restype = C.c_void_p
    lib.uz_vcf_is_bcf.argtypes = [C.c_void_p]
    lib.uz_vcf_header.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.uz_vcf_header.restype = C.c_void_p
    lib.uz_reads_pack_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.uz_reads_pack_exceptions.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
    lib.uz_reads_pack_lists.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.uz_select_qlow_pos_wide.argtypes = [C.c_void_p]
    lib.uz_select_end_derivable.argtypes = [C.c_void_p]
    lib.uz_reads_pack_end_derivable.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
    lib.uz_reads_pack_cigar_omitted.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
    lib.uz_reads_pack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.uz_reads_source_open.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.uz_reads_source_close.argtypes = [C.c_void_p]
    lib.uz_reads_source_close.restype = None
    lib.uz_reads_select_plan.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(C.c_void_p)]
    lib.uz_select_bl_wide.argtypes = [C.c_void_p]
    for fn in (lib.uz_select_n_records, lib.uz_select_n_cigar_total, lib.uz_select_n_row_units, lib.uz_select_n_seq_units, lib.uz_select_n_bl, lib.uz_select_n_bl_units,
               lib.uz_select_n_exc, lib.uz_select_n_qlow_pos, lib.uz_select_n_cigar_omitted, lib.uz_select_n_tuples, lib.uz_select_n_esc16, lib.uz_select_n_esc16_start8, lib.uz_select_n_esc16_narrow8, lib.uz_select_n_esc16_pair8, lib.uz_select_n_new_names):
        fn.argtypes = [C.c_void_p]
        fn.restype = C.c_int64
    lib.uz_reads_select_fill.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.uz_select_pair8_ok.argtypes = [C.c_void_p]
    lib.uz_select_qname_map.argtypes = [C.c_void_p, C.c_void_p]
    lib.uz_select_free.argtypes = [C.c_void_p]
    lib.uz_select_free.restype = None
    lib.uz_index_summary.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int64]
    lib.uz_index_summary.restype = C.c_int64
    lib.uz_bamsrc_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.uz_bamsrc_close.argtypes = [C.c_void_p]
    lib.uz_bamsrc_close.restype = None
    lib.uz_bamsrc_n_contigs.argtypes = [C.c_void_p]
    lib.uz_bamsrc_contig_name.argtypes = [C.c_void_p, C.c_int32]
    lib.uz_bamsrc_contig_name.restype = C.c_char_p
    lib.uz_bamsrc_contig_length.argtypes = [C.c_void_p, C.c_int32]
    lib.uz_bamsrc_tlen_head.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.uz_bamsrc_tlen_head.restype = C.c_int64
    lib.uz_inflate_backend.restype = C.c_char_p
    lib.uz_bam_stage_plan.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]
    lib.uz_bam_stage_begin.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]
    lib.uz_bam_stage_finish.argtypes = [C.c_void_p]
    lib.uz_stage_gather_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.uz_stage_set_inflated.argtypes = [C.c_void_p, C.c_void_p]
    lib.uz_stage_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.uz_stage_sizes.restype = None
    lib.uz_stage_io_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.uz_stage_io_stats.restype = None
    lib.uz_stage_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.uz_stage_timing.restype = None
    lib.uz_stage_fill.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.uz_stage_qname.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
    lib.uz_stage_qname.restype = C.c_void_p
    lib.uz_stage_qnames.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    lib.uz_stage_qnames.restype = C.c_int64
    lib.uz_stage_free.argtypes = [C.c_void_p]
    lib.uz_stage_free.restype = None
    lib.uz_stage_walk_plan_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.uz_stage_walk_plan_sizes.restype = None
    lib.uz_stage_walk_plan.argtypes = [C.c_void_p] * 7
    lib.uz_bam_stage_finish_desc.argtypes = [C.c_void_p] * 5
    lib.uz_stage_kept_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.uz_stage_kept_sizes.restype = None
    lib.uz_stage_kept.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.uz_stage_walk_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.uz_stage_kept_debug.argtypes = [C.c_void_p] * 5
    lib.uz_stage_name_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.uz_packed_block_sums.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.uz_stage_merge_subtasks.argtypes = [C.c_void_p] * 8
    lib.uz_bam_stage_finish_sub.argtypes = [C.c_void_p] * 5
    lib.uz_stage_walk_flagged.argtypes = [C.c_void_p] * 4
    lib.uz_stage_lookup.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.uz_stage_extra.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.uz_stage_n_lookup_tasks.argtypes = [C.c_void_p]
    lib.uz_stage_n_lookup_tasks.restype = C.c_int64
    _LIB = lib
    return lib


class IoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("%s (code %d)" % (msg, code))
        self.code = code


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise IoError(rc, (lib.uz_io_last_error() or b"").decode(errors="replace"))


def read_bam_stream_table(stream: bytes, threads: int = 0, insert_size_max_sample: int = 1000000) -> ReadsTable:
    """uncompressed BAM stream in memory (magic, header, references, records) -> ReadsTable (uz_bam_decode_memory)"""
    lib = load()
    hp = C.c_void_p()
    _check(lib, lib.uz_bam_decode_memory(stream, len(stream), int(threads), C.byref(hp)))
    return _table_from_handle(lib, _Handle(hp.value, lib.uz_bam_free), insert_size_max_sample)


def _arr(ptr, n: int, dtype) -> np.ndarray:
    dtype = np.dtype(dtype)
    if n <= 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class _Handle:
    def __init__(self, ptr, free):
        self.ptr, self._free = ptr, free

    def __del__(self):
        if self.ptr:
            self._free(self.ptr)
            self.ptr = None


class _Names(Sequence):
    """id -> query name, read from the decoded file on demand (no list of 10^8 Python strings)"""

    def __init__(self, lib, handle: _Handle, n: int):
        self._lib, self._h, self._n = lib, handle, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        ln = C.c_int32(0)
        p = self._lib.uz_bam_qname(self._h.ptr, i, C.byref(ln))
        return C.string_at(p, ln.value).decode()


def _table_from_handle(lib, h, insert_size_max_sample: int) -> ReadsTable:
    nc = lib.uz_bam_n_contigs(h.ptr)
    t = ReadsTable([lib.uz_bam_contig_name(h.ptr, i).decode() for i in range(nc)])
    v = abi.ReadsView()
    _check(lib, lib.uz_bam_view(h.ptr, C.byref(v)))
    n = int(v.n_segs)

    def col(name, dtype, count=n):
        return _arr(C.cast(getattr(v, name), C.c_void_p).value, count, dtype)

    t.contig_off = col("contig_off", np.int64, nc + 1).copy()
    t.max_span = col("max_span", np.int32, nc).copy()
    t.start, t.end, t.tlen, t.mate = (col(k, np.int32) for k in ("start", "end", "tlen", "mate"))
    t.flag, t.n_cigar, t.l_seq = (col(k, np.uint16) for k in ("flag", "n_cigar", "l_seq"))
    t.mapq, t.aux = col("mapq", np.uint8), col("aux", np.uint8)
    t.qname, t.cigar_off, t.sq_off16 = (col(k, np.uint32) for k in ("qname", "cigar_off", "sq_off16"))
    t.cigar = col("cigar", np.uint32, int(v.n_cigar_total))
    t.seq = col("seq", np.uint8, int(v.n_sq_bytes))
    t.qual = col("qual", np.uint8, int(v.n_sq_bytes))
    t.qnames = _Names(lib, h, int(v.n_qnames))
    cap = max(int(insert_size_max_sample) + 1, 0)
    k = min(cap, int(lib.uz_bam_n_file_records(h.ptr)))
    head = np.zeros(k, dtype=np.int32)
    if k:
        lib.uz_bam_tlen_head(h.ptr, head.ctypes.data, k)
    t.tlen_head = head
    tm = (C.c_double * 4)()
    lib.uz_bam_timing(h.ptr, tm)
    t.decode_seconds = dict(zip(("read", "inflate", "columns", "names+mates"), (float(x) for x in tm)))
    st = (C.c_int64 * 4)()
    lib.uz_bam_io_stats(h.ptr, st)
    t.io_stats = dict(zip(("file_bytes_read", "blocks_inflated", "records_walked", "records_kept"), (int(x) for x in st)))
    t._native = h  # owns the memory behind the views
    return t


def read_bam_table(path: str, threads: int = 0, insert_size_max_sample: int = 1000000) -> ReadsTable:
    """BAM -> ReadsTable (columns are views into the native handle, kept alive by the table)."""
    lib = load()
    hp = C.c_void_p()
    _check(lib, lib.uz_bam_decode(os.fsencode(path), int(threads), C.byref(hp)))
    return _table_from_handle(lib, _Handle(hp.value, lib.uz_bam_free), insert_size_max_sample)


def bam_index_path(path: str):
    """the BAI next to a BAM (NAME.bam.bai or NAME.bai), or None"""
    for cand in (path + ".bai", path[:-4] + ".bai" if path.endswith(".bam") else None):
        if cand and os.path.isfile(cand):
            return cand
    return None


def read_bam_regions(path: str, tid, lo, hi, threads: int = 0, insert_size_max_sample: int = 1000000, bai: str = None) -> ReadsTable:
    """Index-driven decode: the records the fetches (tid[k], lo[k], hi[k]) return plus, closed under mate(), their mates
    (uz_bam_decode_regions) -- what pysam's fetch() / mate() hand the reference, without inflating the rest of the file."""
    lib = load()
    tid = np.ascontiguousarray(tid, np.int32)
    lo = np.ascontiguousarray(lo, np.int32)
    hi = np.ascontiguousarray(hi, np.int32)
    hp = C.c_void_p()
    _check(lib, lib.uz_bam_decode_regions(os.fsencode(path), os.fsencode(bai) if bai else None, int(tid.size), tid.ctypes.data,
                                          lo.ctypes.data, hi.ctypes.data, int(insert_size_max_sample) + 1, int(threads), C.byref(hp)))
    return _table_from_handle(lib, _Handle(hp.value, lib.uz_bam_free), insert_size_max_sample)


class _Strings(Sequence):
    def __init__(self, fn, handle: _Handle, n: int, post):
        self._fn, self._h, self._n, self._post = fn, handle, n, post

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        ln = C.c_int32(0)
        p = self._fn(self._h.ptr, i, C.byref(ln))
        return self._post(C.string_at(p, ln.value).decode())


def tabix_index_path(path: str):
    """the index next to a sites file -- <path>.tbi next to a BGZF-compressed VCF, else <path>.csi next to that or next to a BCF -- or None"""
    if path.endswith(".gz") and os.path.isfile(path + ".tbi"):
        return path + ".tbi"
    if (path.endswith(".gz") or path.endswith(".bcf")) and os.path.isfile(path + ".csi"):
        return path + ".csi"
    return None


def tabix_contigs(path: str, tbi: str = None) -> list:
    """sequence names of the tabix index, in the order of their first record in the file"""
    lib = load()
    n = lib.uz_vcf_index_names(os.fsencode(path), os.fsencode(tbi) if tbi else None, None, 0)
    if n < 0:
        _check(lib, int(n))
    buf = C.create_string_buffer(max(1, int(n)))
    lib.uz_vcf_index_names(os.fsencode(path), os.fsencode(tbi) if tbi else None, buf, int(n))
    return [x.decode() for x in buf.raw[: int(n)].split(b"\0")[:-1]]


def read_vcf_table_regions(path: str, ref, lo, hi, threads: int = 0, tbi: str = None) -> SitesTable:
    """Index-driven decode: the records overlapping the intervals [lo[k], hi[k]) (0-based) of sequence ref[k] (an index into
    tabix_contigs) -- what cyvcf2's vcf(region) hands the reference, without inflating the rest of the file.  `.io_stats`:
    compressed bytes read, BGZF blocks inflated, lines walked, records kept."""
    lib = load()
    ref = np.ascontiguousarray(ref, np.int32)
    lo = np.ascontiguousarray(lo, np.int32)
    hi = np.ascontiguousarray(hi, np.int32)
    hp = C.c_void_p()
    _check(lib, lib.uz_vcf_decode_regions(os.fsencode(path), os.fsencode(tbi) if tbi else None, int(ref.size), ref.ctypes.data,
                                          lo.ctypes.data, hi.ctypes.data, int(threads), C.byref(hp)))
    t = _vcf_table_from_handle(lib, _Handle(hp.value, lib.uz_vcf_free))
    names = tabix_contigs(path, tbi)
    # utils.py:46-52 get_prefix: the first record of the FILE decides, not the first record of this subset
    t.file_prefix = (names[0][:3] if "chr" in names[0].lower() else "") if names else ""
    return t


def read_vcf_table(path: str, threads: int = 0) -> SitesTable:
    """sites VCF -> SitesTable; `.header` (list of lines) and `.lines[i]` (raw record text) ride along
    for the VCF writer."""
    lib = load()
    hp = C.c_void_p()
    _check(lib, lib.uz_vcf_decode(os.fsencode(path), int(threads), C.byref(hp)))
    return _vcf_table_from_handle(lib, _Handle(hp.value, lib.uz_vcf_free))


def _vcf_table_from_handle(lib, h) -> SitesTable:
    v = VcfView()
    _check(lib, lib.uz_vcf_view_get(h.ptr, C.byref(v)))
    n, ns, nc = int(v.n_sites), int(v.n_samples), int(v.n_contigs)
    t = SitesTable([lib.uz_vcf_sample(h.ptr, i).decode() for i in range(ns)],
                   [lib.uz_vcf_contig(h.ptr, i).decode() for i in range(nc)])
    t.contig_off = _arr(v.contig_off, nc + 1, np.int64).copy()
    t.pos, t.end = _arr(v.pos, n, np.int32), _arr(v.end, n, np.int32)
    t.sflags, t.ref_base, t.alt_base = (_arr(getattr(v, k), n, np.uint8) for k in ("sflags", "ref_base", "alt_base"))
    t.gt = _arr(v.gt, ns * n, np.uint8).reshape(ns, n)
    t.ref_depth = _arr(v.ref_depth, ns * n, np.int32).reshape(ns, n)
    t.alt_depth = _arr(v.alt_depth, ns * n, np.int32).reshape(ns, n)
    t.gq = _arr(v.gq, ns * n, np.float64).reshape(ns, n)
    t.ref_str = _Strings(lib.uz_vcf_ref, h, n, lambda s: s)
    t.alt_strs = _Strings(lib.uz_vcf_alt, h, n, lambda s: [] if s == "." else s.split(","))
    t.lines = _Strings(lib.uz_vcf_line, h, n, lambda s: s)
    ln = C.c_int64(0)
    hp2 = lib.uz_vcf_header(h.ptr, C.byref(ln))
    t.header = C.string_at(hp2, ln.value).decode().split("\n") if ln.value else []
    t.is_bcf = bool(lib.uz_vcf_is_bcf(h.ptr))

    def info(i: int, key: str):
        ln2 = C.c_int32(0)
        p = lib.uz_vcf_info(h.ptr, int(i), key.encode(), C.byref(ln2))
        return C.string_at(p, ln2.value).decode() if p else None

    t.info = info
    st = (C.c_int64 * 4)()
    lib.uz_vcf_io_stats(h.ptr, st)
    t.io_stats = [int(x) for x in st]
    t._native = h
    return t


# ---------------------------------------------------------------------------- staged (packed) records
def pack_reads(reads: "abi.Held", min_base_qual: int, threads: int = 0, alloc=None, two_bit: bool = True, lists: bool = True,
               with_end=True, cigar_compact: bool = False) -> "abi.Held":
    """ASCII table (abi.reads_view / a decoder's view) -> the packed form uz_reads_upload_packed takes, for the
    base-quality threshold of the run.  alloc(nbytes) -> uint8 array chooses the memory (pinned for the upload).
    two_bit: base rows in two bits + the listed bases that are not A/C/G/T (half the bytes of the largest column on the
    host link; the device expands them); False = BAM's four-bit codes.
    lists: the quality plane as per-record counts + the positions of the records that can need them (a quarter of the plane's
    bytes; the device rebuilds the rows) -- for tables that serve batches of point variants only (uz_types.h); False = the plane
    itself.
    with_end: True keeps the `end` column (a table that will be the source of selections needs it); None = leave it out when every
    record's end is what its CIGAR gives (a BAM decoder's table) -- the device derives it.
    cigar_compact: the CIGAR word of a record that is one M / = / X over the read stays home (its aux byte names the operation; not
    for a table that will be the source of selections)."""
    lib = load()
    nc, nu = C.c_int64(0), C.c_int64(0)
    _check(lib, lib.uz_reads_pack_sizes(reads.ref(), C.byref(nc), C.byref(nu)))
    n_exc = None
    if two_bit:
        ne = C.c_int64(0)
        _check(lib, lib.uz_reads_pack_exceptions(reads.ref(), int(threads), C.byref(ne)))
        n_exc = int(ne.value)
    n_qpos, wide = None, False
    if lists:
        nq, wd = C.c_int64(0), C.c_int32(0)
        _check(lib, lib.uz_reads_pack_lists(reads.ref(), int(min_base_qual), int(threads), C.byref(nq), C.byref(wd)))
        n_qpos, wide = int(nq.value), bool(wd.value)
    if with_end is None:
        yes = C.c_int32(0)
        _check(lib, lib.uz_reads_pack_end_derivable(reads.ref(), int(threads), C.byref(yes)))
        with_end = not yes.value
    omitted = None
    if cigar_compact:
        om = C.c_int64(0)
        _check(lib, lib.uz_reads_pack_cigar_omitted(reads.ref(), int(threads), C.byref(om)))
        omitted = int(om.value)
    out = abi.packed_view_alloc(int(reads.view.n_segs), int(reads.view.n_contigs), nc.value, nu.value, alloc, n_exc=n_exc,
                                n_qlow_pos=n_qpos, qlow_pos_wide=wide, with_end=bool(with_end), cigar_omitted=omitted)
    _check(lib, lib.uz_reads_pack(reads.ref(), int(min_base_qual), int(threads), out.ref()))
    return out


class ReadsSource:
    """A packed table in host memory opened for fetch-reach selections (uz_reads_source_open)."""

    def __init__(self, packed: "abi.Held", threads: int = 0):
        self.lib = load()
        self.packed = packed  # keeps the arrays alive
        h = C.c_void_p()
        _check(self.lib, self.lib.uz_reads_source_open(packed.ref(), int(threads), C.byref(h)))
        self._h = _Handle(h, self.lib.uz_reads_source_close)
        self.threads = threads

    def select(self, contig, lo, hi, alloc=None, want_index=False, all_bases=False, lists=True, with_end=None, extra=None, cigar_compact=True,
               tuples=True, d16=True, start8=True, wide_no_units=False, narrow8=True, pair8=True, base_lists=None, tup8=None):
        """The records the fetches (contig[k], lo[k], hi[k]) return plus their mates, as a packed table.
        tup8 (default: on, UZ_TUP8=0 turns it off): the dictionary index in one byte (uz_types.h tup8; abi.compact_tup) -- what crosses the link.  Records that
        are reachable only as mates are staged without their bases unless all_bases (--no-extended batches need them).
        lists: the qualities as counts + listed positions (what the host link carries); False keeps the plane (plane sources only).
        extra (uint16 per fetch, staging.fetch_points(..., allele_len=)): stage only the 32-base units of a record's rows that
        the read stage can read -- the fetched position and `extra` bases on (needs lists; ignored with all_bases).
        tuples: flag, l_seq, n_cigar, mapq, aux (and n_low) as a 16-bit index into a table of their combinations (falls back to the
        plain columns when the selection holds more than 65536 of them).
        d16: start, tlen, mate and qname as 16-bit differences (to the record before / to the record's own index) + an escape list.
        start8: ... the start differences in eight bits (records of a pile-up start a few bases apart).
        wide_no_units (SV batches, with extra): a fetch wider than two bases stages no unit of the records it returns.
        narrow8 (with start8): ... and the mate / name-id differences in eight bits (a selection keeps a third of a pile-up: a mate lies
        within +-75 kept records).
        pair8 (with start8): tlen, mate and name id in the pair form's ONE byte (uz_types.h, pair_d8) when the source's name ids ascend
        by first appearance (any decoder's table; else the narrow8 form).  The output then numbers the names of the selection by first
        appearance: `.qname_map[output id]` = the source's id."""
        contig = np.ascontiguousarray(contig, np.int32)
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        if tup8 is None:
            # (not for an SV batch: its step is a chain of dependent launches over three heavy chunks, not bytes -- the kernel that rebuilds the
            # 16-bit column stands in that chain: config 5 4.56 ms with the 16-bit index, 4.64 with the byte)
            tup8 = os.environ.get("UZ_TUP8", "1") != "0" and not wide_no_units
        sel = C.c_void_p()
        masks = extra is not None and lists and not all_bases
        if masks:
            extra = np.ascontiguousarray(extra, np.uint16)
            assert extra.size == contig.size
        _check(self.lib, self.lib.uz_reads_select_plan(self._h.ptr, int(contig.size), contig.ctypes.data, lo.ctypes.data,
                                                       hi.ctypes.data, 1 if all_bases else 0,
                                                       ((3 if wide_no_units else 1) | (4 if (os.environ.get("UZ_BASE_LISTS", "1") != "0" if base_lists is None else base_lists) else 0)) if masks else 0,
                                                       extra.ctypes.data if masks else None,
                                                       (1 | (2 if cigar_compact else 0) | (4 if lists else 0)) if tuples else 0,
                                                       int(self.threads), C.byref(sel)))
        try:
            n = self.lib.uz_select_n_records(sel)
            two_bit = bool(self.packed.view.seq2)  # a selection keeps the base-row form of its source
            n_tup = int(self.lib.uz_select_n_tuples(sel)) if tuples else -1
            n_bl = int(self.lib.uz_select_n_bl(sel))
            pair8 = bool(pair8 and d16 and start8 and self.lib.uz_select_pair8_ok(sel))
            narrow8 = bool(narrow8 and not pair8)
            out = abi.packed_view_alloc(n, int(self.packed.view.n_contigs), self.lib.uz_select_n_cigar_total(sel),
                                        self.lib.uz_select_n_row_units(sel), alloc, n_seq_units=self.lib.uz_select_n_seq_units(sel),
                                        n_exc=int(self.lib.uz_select_n_exc(sel)) if two_bit else None,
                                        n_qlow_pos=int(self.lib.uz_select_n_qlow_pos(sel)) if lists else None,
                                        qlow_pos_wide=bool(self.lib.uz_select_qlow_pos_wide(sel)) if lists else False,
                                        with_end=(not self.lib.uz_select_end_derivable(sel)) if with_end is None else bool(with_end),
                                        with_umask=masks, cigar_omitted=int(self.lib.uz_select_n_cigar_omitted(sel)) if cigar_compact else None,
                                        n_tup=n_tup if n_tup >= 0 else None,
                                        n_esc16=int((self.lib.uz_select_n_esc16_pair8 if pair8 else self.lib.uz_select_n_esc16_narrow8 if (start8 and narrow8)
                                                     else self.lib.uz_select_n_esc16_start8 if start8 else self.lib.uz_select_n_esc16)(sel)) if d16 else None,
                                        start8=bool(d16 and start8), narrow8=bool(d16 and start8 and narrow8), pair8=pair8,
                                        n_bl=n_bl if n_bl >= 0 else None, n_bl_units=int(self.lib.uz_select_n_bl_units(sel)), bl_wide=bool(self.lib.uz_select_bl_wide(sel)),
                                        pk_sums=_PK_SUMS, tup_off_link=bool(tup8 and n_tup >= 0))
            idx = np.zeros(max(1, n), np.int32) if want_index else None
            _check(self.lib, self.lib.uz_reads_select_fill(sel, int(self.threads), out.ref(),
                                                           idx.ctypes.data if want_index else None))
            block_sums(out, self.threads)  # (the span sums the device's header build packs from: uz_types.h pk_sums)
            if tup8 and n_tup >= 0:
                abi.compact_tup(out, alloc)
            out.qname_map = None
            if pair8:
                out.qname_map = np.zeros(max(1, int(self.lib.uz_select_n_new_names(sel))), np.uint32)
                _check(self.lib, self.lib.uz_select_qname_map(sel, out.qname_map.ctypes.data))
                out.qname_map = out.qname_map[: int(self.lib.uz_select_n_new_names(sel))]
        finally:
            self.lib.uz_select_free(sel)
        return (out, idx[:n]) if want_index else out


# ---------------------------------------------------------------------------- BAM file -> staged records in one pass
STAGE_ALL_BASES, STAGE_UNIT_MASKS, STAGE_PLANE, STAGE_WIDE_NO_UNITS, STAGE_BASE_LISTS, STAGE_SMALL_TASKS = 1, 2, 4, 8, 16, 32


def index_summary(path: str, kind: str) -> np.ndarray:
    """what the native BAI / TBI reader makes of an index file (uz_index_summary): int64 [n_ref, 6]"""
    lib = load()
    k = {"bai": 0, "tbi": 1}[kind]
    n = lib.uz_index_summary(os.fsencode(path), k, None, 0)
    if n < 0:
        _check(lib, int(n))
    out = np.zeros((max(1, int(n)), 6), np.int64)
    lib.uz_index_summary(os.fsencode(path), k, out.ctypes.data, int(n))
    return out[: int(n)]


def inflate_backend() -> str:
    return load().uz_inflate_backend().decode()


def default_threads() -> int:
    """worker threads the library uses for `threads=0`: CPUs of the affinity mask, held to twice the cgroup's CPU quota"""
    return int(load().uz_io_default_threads())


def cpu_quota() -> int:
    """CPUs the container's cgroup grants (cpu.max); 0 = no limit"""
    return int(load().uz_io_cpu_quota())


class _StageNames(Sequence):
    """id -> query name of a staged batch (uz_stage_qname)"""

    def __init__(self, lib, handle: _Handle, n: int):
        self._lib, self._h, self._n = lib, handle, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        ln = C.c_int32(0)
        p = self._lib.uz_stage_qname(self._h.ptr, i, C.byref(ln))
        return C.string_at(p, ln.value).decode()

    def take(self, ids) -> list:
        """the names of many ids in one call (uz_stage_qnames): a batch's result lists name hundreds of thousands of reads"""
        ids = np.ascontiguousarray(ids, np.uint32)
        n = int(ids.size)
        if n == 0:
            return []
        off = np.zeros(n + 1, np.int64)
        need = int(self._lib.uz_stage_qnames(self._h.ptr, ids.ctypes.data, n, None, 0, off.ctypes.data))
        if need < 0:
            raise IndexError("query-name id out of range")
        buf = np.empty(max(1, need), np.uint8)
        self._lib.uz_stage_qnames(self._h.ptr, ids.ctypes.data, n, buf.ctypes.data, need, None)
        text = buf[:need].tobytes().decode()
        o = off.tolist()
        return [text[o[k]: o[k + 1]] for k in range(n)]


# include/uz_bamwalk.h
WALK_DESC = np.dtype([("voff", "<u8"), ("src", "<u8"), ("h1", "<u8"), ("pos", "<i4"), ("end", "<i4"), ("tlen", "<i4"), ("mpos", "<i4"), ("mtid", "<i4"),
                      ("h2", "<u4"), ("task", "<u4"), ("flag", "<u2"), ("l_seq", "<u2"), ("n_cigar", "<u2"), ("mapq", "u1"), ("l_name", "u1"), ("direct", "u1"),
                      ("pad8", "u1"), ("pad16", "<u2")])
KEPT_REC = np.dtype([("src", "<u8"), ("qname", "<u4"), ("mate", "<i4"), ("cig_off", "<u4"), ("unit_off", "<u4"), ("seq_off", "<u4"), ("name_off", "<u4")])
NEED_REC = np.dtype([("h1", "<u8"), ("h2", "<u4"), ("l_name", "<u4"), ("mtid", "<i4"), ("mpos", "<i4"), ("who", "<u4"), ("pad", "<u4")])
assert WALK_DESC.itemsize == 64 and KEPT_REC.itemsize == 32 and NEED_REC.itemsize == 32
WALK_TASK_JOIN = 0x80000000
WALK_TASK_COLS, WALK_SPAN_COLS = 10, 6
KEPT_NO_SEQ = 0xFFFFFFFF
WALK_SRC_AUX = 1 << 63


_PK_SUMS = os.environ.get("UZ_PK_SUMS", "1") != "0"  # the packers hand the device the span sums of what they packed (UZ_PK_SUMS=0: the device computes them)


def block_sums(held: "abi.Held", threads: int = 0) -> "abi.Held":
    """The span sums of a filled packed view (uz_packed_block_sums) into its pk_sums array (abi.packed_view_alloc(pk_sums=True)): the device's header
    build then packs from them instead of computing them in a pass of its own.  A view without the array is returned as it is."""
    if "pk_sums" not in held.arrays:
        return held
    lib = load()
    nb = C.c_int64(0)
    _check(lib, lib.uz_packed_block_sums(held.ref(), int(threads), held.arrays["pk_sums"].ctypes.data, C.byref(nb)))
    assert int(nb.value) == int(held.view.n_pk_spans)
    return held


def stage_kept_debug(lib, stage_ptr, n: int):
    """(voff, qname, mate, bases) of the kept records of a finished plan (uz_stage_kept_debug)"""
    voff, qn, mt, bs = np.zeros(max(1, n), np.uint64), np.zeros(max(1, n), np.uint32), np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.uint8)
    _check(lib, lib.uz_stage_kept_debug(stage_ptr, voff.ctypes.data, qn.ctypes.data, mt.ctypes.data, bs.ctypes.data))
    return voff[:n], qn[:n], mt[:n], bs[:n]


class _KeptNames(Sequence):
    """name id -> query name of a batch walked on the device: the kept records' names came back from HBM in record order (uz_reads_from_bam), and
    the id's name is the name of the record that brought it first (uz_stage_name_records: asked per look-up, for the ids looked up -- a chunk's
    result lists name a few per cent of its million names)"""

    def __init__(self, buf: np.ndarray, name_off: np.ndarray, total: int, n_names: int, stage):
        self._buf, self._off, self._total, self._n, self._stage = buf, name_off, int(total), int(n_names), stage

    def __len__(self):
        return self._n

    def _records(self, ids: np.ndarray) -> np.ndarray:
        lib = load()
        ids = np.ascontiguousarray(ids, np.uint32)
        rec = np.zeros(max(1, ids.size), np.int64)
        _check(lib, lib.uz_stage_name_records(self._stage.ptr, ids.ctypes.data, int(ids.size), rec.ctypes.data))
        return rec[: ids.size]

    def _bounds(self, r: np.ndarray):
        n = self._off.shape[0]
        a = self._off[r].astype(np.int64)
        nxt = r + 1
        b = np.where(nxt < n, self._off[np.minimum(nxt, n - 1)].astype(np.int64), self._total)
        return a, b

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        a, b = self._bounds(self._records(np.array([int(i)], np.int64)))
        return self._buf[int(a[0]): int(b[0])].tobytes().decode()

    def take(self, ids) -> list:
        a, b = self._bounds(self._records(np.asarray(ids)))
        mv = memoryview(self._buf)  # (not tobytes(): the buffer holds the names of every kept record of the batch, tens of megabytes)
        return [str(mv[x:y], "utf-8") for x, y in zip(a.tolist(), b.tolist())]


def names_of_buffer(buf: np.ndarray, off: np.ndarray) -> list:
    """read names lying back to back in `buf` (off [n + 1]) as strings: the buffer decoded ONCE and cut (a SAM read name is printable ASCII, so
    byte offsets are character offsets; half the time of a decode per name: 150 k names in 32 ms instead of 58); anything else, name by name"""
    o = off.tolist()
    try:
        text = buf[: o[-1]].tobytes().decode("ascii")
    except UnicodeDecodeError:
        mv = memoryview(buf)
        return [str(mv[x:y], "utf-8") for x, y in zip(o[:-1], o[1:])]
    return [text[x:y] for x, y in zip(o[:-1], o[1:])]


class DeviceNames(Sequence):
    """name id -> query name of a batch whose joins ran on the device: the names stay in HBM with the table (uz_reads_from_walk) and the ids a
    chunk's result lists name -- a few per cent of its names -- are answered by uz_reads_names.  Valid while the table lives."""

    def __init__(self, engine, rid: int, n_names: int):
        self._eng, self._rid, self._n = engine, int(rid), int(n_names)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.take(np.arange(*i.indices(self._n), dtype=np.uint32))
        return self._eng.reads_names(self._rid, np.array([int(i)], np.uint32))[0]

    def take(self, ids) -> list:
        return self._eng.reads_names(self._rid, np.asarray(ids))

    def frozen(self) -> "_FrozenNames":
        """every name, on the host: for a caller that frees the table before it asks"""
        buf, off = self._eng.reads_names_raw(self._rid, np.arange(self._n, dtype=np.uint32))
        return _FrozenNames(buf, off)


class _FrozenNames(Sequence):
    def __init__(self, buf: np.ndarray, off: np.ndarray):
        self._buf, self._off = buf, off

    def __len__(self):
        return int(self._off.size) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        i = int(i)
        return self._buf[int(self._off[i]): int(self._off[i + 1])].tobytes().decode()

    def take(self, ids) -> list:
        ids = np.asarray(ids, np.int64)
        mv = memoryview(self._buf)
        return [str(mv[x:y], "utf-8") for x, y in zip(self._off[ids].tolist(), self._off[ids + 1].tolist())]


class KeptBatch:
    """What BamSource.select_kept returns: the records fetch() + mate() hand the reference for a batch, as the list the device packs its record
    table from (uz_kept_rec), with the table's sizes.  `.plan`: the gathered BGZF blocks and the walk plan the device worked from.
    joined: the joins ran on the device too (select_kept(join=engine)) -- the list lies in the batch's slot in HBM, `.kept` is None."""
    qnames = None
    token = None
    _release = None
    joined = False

    def __del__(self):  # (dropped before its table was built -- a failed join, an abandoned pipeline: the inflated bytes on the device are given up)
        if self.token is not None and self._release is not None:
            try:
                self._release(self.token)
            except Exception:
                pass
            self.token = None

    def set_names(self, buf: np.ndarray):
        self.qnames = _KeptNames(buf, self.kept["name_off"], int(self.n_name_bytes), int(self.n_qnames), self._stage)


def _stage_extra(lib, sh, tot, d0, a0):
    """what the stage holds for the device from descriptor d0 / aux byte a0 on (uz_stage_extra) -> (descriptors, aux bytes, reference of every look-up task)"""
    nd, na = int(tot[0]) - d0, int(tot[1]) - a0
    x, aux = np.zeros(max(1, nd), WALK_DESC), np.zeros(max(1, na), np.uint8)
    nl = int(lib.uz_stage_n_lookup_tasks(sh.ptr))
    lt = np.zeros(max(1, nl), np.int32)
    _check(lib, lib.uz_stage_extra(sh.ptr, d0, a0, x.ctypes.data, aux.ctypes.data, lt.ctypes.data))
    return x[:nd], aux[:na], lt[:nl]


class BamSource:
    """An indexed BAM opened for staging (uz_bamsrc_open): `select` turns the fetches of one batch straight into the packed table
    uz_reads_upload_packed takes -- what read_bam_regions + pack_reads + ReadsSource.select build in three passes, in one."""

    def __init__(self, path: str, bai: str = None, insert_size_max_sample: int = 1000000, threads: int = 0):
        self.lib = load()
        h = C.c_void_p()
        _check(self.lib, self.lib.uz_bamsrc_open(os.fsencode(path), os.fsencode(bai) if bai else None, int(insert_size_max_sample) + 1, C.byref(h)))
        self._h = _Handle(h.value, self.lib.uz_bamsrc_close)
        self.path, self.threads = path, threads
        nc = self.lib.uz_bamsrc_n_contigs(h)
        self.contigs = [self.lib.uz_bamsrc_contig_name(h, i).decode() for i in range(nc)]
        self.contig_len = [self.lib.uz_bamsrc_contig_length(h, i) for i in range(nc)]
        head = np.zeros(int(insert_size_max_sample) + 1, np.int32)
        k = self.lib.uz_bamsrc_tlen_head(h, head.ctypes.data, head.size)
        self.tlen_head = head[: int(k)].copy()

    def select(self, contig, lo, hi, min_base_qual: int, alloc=None, all_bases=False, lists=True, extra=None, pool=None, wide_no_units=False,
               inflate=None, inflate_alloc=None, inflate_max_bytes=16 << 30, base_lists=None, tup8=None):
        """-> abi.Held packed view (the same columns, byte for byte, as ReadsSource.select's defaults on the region-decoded table).
        `.io_stats` / `.timing` / `.qnames` ride on the returned object.
        pool (engine.PinnedPool): the columns are carved from ONE page-locked block of exactly the planned size (they cross the link
        as one copy); alloc: any other allocator of the columns.
        inflate(comp, comp_bytes, in_off, out_off, out): somebody who inflates BGZF blocks faster than this host's cores
        (HipEngine.inflate_blocks: the device) -- the blocks the walk will read are gathered and handed to it between the two halves of the
        plan, and the walk copies records out of what comes back (each block still held against its CRC-32) instead of inflating;
        inflate_alloc(nbytes) -> uint8 array: where the gathered and the inflated bytes go (pinned memory for full link speed); a batch whose
        blocks inflate to more than inflate_max_bytes stays with the host's inflate (that much memory would have to be page-locked)."""
        if tup8 is None:
            tup8 = os.environ.get("UZ_TUP8", "1") != "0" and not wide_no_units  # (as ReadsSource.select)
        contig = np.ascontiguousarray(contig, np.int32)
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        masks = extra is not None and lists and not all_bases
        if extra is not None:
            extra = np.ascontiguousarray(extra, np.uint16)
            assert extra.size == contig.size
        if base_lists is None:  # (UZ_BASE_LISTS=0: every record's staged units as rows, the form of round 3)
            base_lists = os.environ.get("UZ_BASE_LISTS", "1") != "0"
        flags = (STAGE_ALL_BASES if all_bases else 0) | (STAGE_UNIT_MASKS if masks else 0) | (0 if lists else STAGE_PLANE) | (STAGE_WIDE_NO_UNITS if wide_no_units and masks else 0) \
            | (STAGE_BASE_LISTS if base_lists and masks else 0)
        st = C.c_void_p()
        pre = None
        if inflate is None:
            _check(self.lib, self.lib.uz_bam_stage_plan(self._h.ptr, int(contig.size), contig.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                                        extra.ctypes.data if extra is not None else None, flags, int(min_base_qual), int(self.threads),
                                                        C.byref(st)))
            sh = _Handle(st.value, self.lib.uz_stage_free)
        else:
            _check(self.lib, self.lib.uz_bam_stage_begin(self._h.ptr, int(contig.size), contig.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                                         extra.ctypes.data if extra is not None else None, flags, int(min_base_qual), int(self.threads),
                                                         C.byref(st)))
            sh = _Handle(st.value, self.lib.uz_stage_free)
            nb, cb, ob = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            _check(self.lib, self.lib.uz_stage_gather_blocks(sh.ptr, None, 0, None, None, C.byref(nb), C.byref(cb), C.byref(ob)))
            if nb.value and ob.value <= int(inflate_max_bytes):
                ia = inflate_alloc or (lambda nbytes: np.empty(max(16, nbytes), np.uint8))
                comp = ia(int(cb.value) + 64)
                inflated = ia(int(ob.value) + 64)
                in_off, out_off = np.zeros(nb.value, np.int64), np.zeros(nb.value + 1, np.int64)
                t_g = time.perf_counter()
                _check(self.lib, self.lib.uz_stage_gather_blocks(sh.ptr, comp.ctypes.data, int(cb.value), in_off.ctypes.data, out_off.ctypes.data, None, None, None))
                t_i = time.perf_counter()
                inflate(comp, int(cb.value), in_off, out_off, inflated)
                t_e = time.perf_counter()
                _check(self.lib, self.lib.uz_stage_set_inflated(sh.ptr, inflated.ctypes.data))
                pre = dict(blocks=int(nb.value), comp_bytes=int(cb.value), out_bytes=int(ob.value), gather_s=t_i - t_g, inflate_s=t_e - t_i, keep=(comp, inflated))
            _check(self.lib, self.lib.uz_bam_stage_finish(sh.ptr))
        z = (C.c_int64 * 16)()
        self.lib.uz_stage_sizes(sh.ptr, z)
        n, n_cig, n_om, n_units, n_seq, n_exc, n_qpos, wide, n_tup, n_esc, n_names, has_um, has_bl, n_bl, n_blu = (int(x) for x in z[:15])
        if pool is not None:  # the sizes are known now: one block for all the columns (each 256-byte aligned)
            total = (n * 5 + n_cig * 4 + n_seq * 8 + n_exc * 7 + n_qpos * 2 + n_esc * 12 + n_tup * 13 + len(self.contigs) * 12 + (0 if lists else n_units * 4)
                     + n_bl * 3 + 44 * 256 + 4096 + (abi.pk_spans(n) + 1) * 8 * abi.PK_SUMS + 256)
            if not (getattr(pool, "keep", False) and pool.rewind(total)):  # (a kept pool: its block is re-used when it is large enough)
                if getattr(pool, "keep", False):
                    pool.free_all()
                    total += total // 4
                pool.new_slab(total)
            alloc = pool.alloc
        out = abi.packed_view_alloc(n, len(self.contigs), n_cig + n_om, n_units, alloc, n_seq_units=n_seq, n_exc=n_exc,
                                    n_qlow_pos=n_qpos if lists else None, qlow_pos_wide=bool(wide) and lists, with_end=False, with_umask=bool(has_um),
                                    cigar_omitted=n_om, n_tup=n_tup, n_esc16=n_esc, start8=True, pair8=True,
                                    n_bl=n_bl if has_bl else None, n_bl_units=n_blu, bl_wide=bool(wide) and bool(has_bl), pk_sums=_PK_SUMS,
                                    tup_off_link=bool(tup8))
        _check(self.lib, self.lib.uz_stage_fill(sh.ptr, int(self.threads), out.ref()))
        block_sums(out, self.threads)  # (the span sums the device's header build packs from: uz_types.h pk_sums)
        if tup8:
            abi.compact_tup(out, alloc)  # (the dictionary index in one byte: uz_types.h tup8)
        io = (C.c_int64 * 8)()
        self.lib.uz_stage_io_stats(sh.ptr, io)
        tm = (C.c_double * 6)()
        self.lib.uz_stage_timing(sh.ptr, tm)
        out.io_stats = dict(zip(("file_bytes_read", "blocks_inflated", "records_walked", "records_kept", "reach_intervals", "index_mate_lookups",
                                 "blocks_from_the_device", "gathered_bytes"), (int(x) for x in io)))
        out.pre_inflate = None if pre is None else {k: v for k, v in pre.items() if k != "keep"}
        out.timing = dict(zip(("spans", "walk", "mates", "numbering", "fill"), (float(x) for x in tm)))
        out.qnames = _StageNames(self.lib, sh, n_names)
        out._stage = sh
        return out

    def _joined_on_the_device(self, dev, sh, plan, n_host, nt, min_base_qual, all_bases, release, t0, t1) -> "KeptBatch":
        """select_kept(join=dev): walk AND joins on the device (dev: HipEngine -- walk / walk_flags / join / join_needs).  Nothing of the batch's
        records comes down the link: up go the compressed blocks and the plan, down come the walk tasks' flags, the totals -- and, when the file has
        them, the short list of mates only the index can answer (uz_stage_lookup), whose records go up as descriptors + bytes."""
        lib = self.lib
        token, n_desc = dev.walk(plan)
        t2 = time.perf_counter()
        try:
            d_flags, d_walked = dev.walk_flags(token, nt)
            if os.environ.get("UZ_TEST_FLAG_EVERY"):  # test hook: every n-th walk task counts as handed back by the device -- the host walks its stage task
                d_flags = d_flags.copy()
                d_flags[:: max(1, int(os.environ["UZ_TEST_FLAG_EVERY"]))] |= 1
            h_flags = np.zeros(max(1, n_host), np.int32)
            tot = (C.c_int64 * 2)()
            _check(lib, lib.uz_stage_walk_flagged(sh.ptr, np.ascontiguousarray(d_flags, np.int32).ctypes.data if nt else None, h_flags.ctypes.data, tot))
            d0 = a0 = 0
            x, aux, lt = _stage_extra(lib, sh, tot, d0, a0)
            d0, a0 = int(tot[0]), int(tot[1])
            need_ans, rounds, n_lookups = None, 0, 0
            while True:
                n_need, totals = dev.join(token, n_host, h_flags, len(self.contigs), all_bases, x, aux, lt, need_ans)
                rounds += 1
                if n_need == 0:
                    break
                if rounds > 64:
                    raise IoError(-4, "the mate closure of the batch did not end after 64 trips through the index")
                need = dev.join_needs(token, n_need)
                need_ans = np.zeros(n_need, np.int32)
                _check(lib, lib.uz_stage_lookup(sh.ptr, int(n_need), need.ctypes.data, need_ans.ctypes.data, tot))
                n_lookups += n_need
                x, aux, lt = _stage_extra(lib, sh, tot, d0, a0)
                d0, a0 = int(tot[0]), int(tot[1])
        except BaseException:
            if release is not None:
                release(token)
            raise
        t3 = time.perf_counter()
        out = KeptBatch()
        out.joined, out.token, out._release = True, token, release
        out.n, out.n_qnames, out.n_cigar_total, out.n_row_units, out.n_seq_units, out.n_name_bytes = (int(v) for v in totals[:6])
        out.n_aux, out.host_tasks, out.kept = a0, int(np.count_nonzero(h_flags[:n_host])), None
        out.n_desc, out.n_extra_desc, out.join_calls = int(n_desc), d0, rounds
        io = (C.c_int64 * 8)()
        lib.uz_stage_io_stats(sh.ptr, io)
        out.io_stats = dict(zip(("file_bytes_read", "blocks_inflated", "records_walked", "records_kept", "reach_intervals", "index_mate_lookups",
                                 "blocks_from_the_device", "gathered_bytes"), (int(v) for v in io)))
        out.io_stats["records_walked"] += int(d_walked.sum())
        out.io_stats["records_kept"] = out.n
        out.timing = dict(plan=t1 - t0, walk=t2 - t1, joins=t3 - t2, kept=0.0, mates=0.0, numbering=0.0)
        out.plan, out.desc, out.d_first, out.d_flags = plan, None, None, d_flags
        out.min_base_qual, out.n_contigs, out.all_bases = int(min_base_qual), len(self.contigs), bool(all_bases)
        out._stage = sh
        return out

    def select_kept(self, contig, lo, hi, min_base_qual: int, walk=None, all_bases=False, alloc=None, extra=None, release=None, small_tasks=None, merge=False,
                    join=None) -> "KeptBatch":
        """The same batch through the device's walk (include/uz_bamwalk.h): the blocks are gathered, `walk(plan)` inflates them in HBM and walks
        them there (HipEngine.bam_walk -> descriptors, d_first, d_flags, d_walked; None: the host's twin uz_stage_walk_host -- tests), the
        batch-wide joins run here on the descriptors, and the answer is the list of kept records for uz_reads_from_bam.
        plan: dict(comp, comp_bytes, in_off, out_off, out_bytes, task, span, reach, fetch, blk_coff).
        merge: the descriptors of the plan's sub-tasks are joined per task of the stage first (uz_stage_merge_subtasks: `.desc` / `.d_first` are then per
        task, for the parity tests); default: the joins read them as they came (uz_bam_stage_finish_sub).
        release(token): gives the walked batch on the device up (HipEngine.bam_walk_release) -- called when the joins fail here, or when the returned
        batch is dropped without its table having been built."""
        contig = np.ascontiguousarray(contig, np.int32)
        lo = np.ascontiguousarray(lo, np.int32)
        hi = np.ascontiguousarray(hi, np.int32)
        if extra is not None:
            extra = np.ascontiguousarray(extra, np.uint16)
        ia = alloc or (lambda nbytes: np.empty(max(16, nbytes), np.uint8))
        if small_tasks is None:  # (a plan for the device's walk takes smaller tasks: one wavefront walks a task)
            small_tasks = (walk is not None or join is not None) and os.environ.get("UZ_STAGE_SUBTASKS", "1") != "0"  # (0: the stage's own tasks as walk tasks -- a development aid)
        flags = (STAGE_ALL_BASES if all_bases else 0) | (STAGE_SMALL_TASKS if small_tasks else 0)
        st = C.c_void_p()
        t0 = time.perf_counter()
        _check(self.lib, self.lib.uz_bam_stage_begin(self._h.ptr, int(contig.size), contig.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                                     extra.ctypes.data if extra is not None else None, flags, int(min_base_qual), int(self.threads), C.byref(st)))
        sh = _Handle(st.value, self.lib.uz_stage_free)
        nb, cb, ob = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(self.lib, self.lib.uz_stage_gather_blocks(sh.ptr, None, 0, None, None, C.byref(nb), C.byref(cb), C.byref(ob)))
        comp = ia(int(cb.value) + 64)
        in_off, out_off = np.zeros(max(1, nb.value), np.int64), np.zeros(nb.value + 1, np.int64)
        if nb.value:
            _check(self.lib, self.lib.uz_stage_gather_blocks(sh.ptr, comp.ctypes.data, int(cb.value), in_off.ctypes.data, out_off.ctypes.data, None, None, None))
        z = (C.c_int64 * 8)()
        self.lib.uz_stage_walk_plan_sizes(sh.ptr, z)
        nt, nsp, nr, nf, nblk, n_host = (int(x) for x in z[:6])  # (nt: the device's walk tasks -- sub-tasks of the stage's n_host tasks under small_tasks)
        task = np.zeros((max(1, nt), WALK_TASK_COLS), np.int32)
        span = np.zeros((max(1, nsp), WALK_SPAN_COLS), np.int64)
        reach = np.zeros((max(1, nr), 2), np.int32)
        fetch = np.zeros((max(1, nf), 3), np.int32)
        blk_coff, blk_crc = np.zeros(max(1, nblk), np.int64), np.zeros(max(1, nblk), np.uint32)
        _check(self.lib, self.lib.uz_stage_walk_plan(sh.ptr, task.ctypes.data, span.ctypes.data, reach.ctypes.data, fetch.ctypes.data, blk_coff.ctypes.data,
                                                     blk_crc.ctypes.data))
        plan = dict(comp=comp, comp_bytes=int(cb.value), in_off=in_off[: nb.value], out_off=out_off, out_bytes=int(ob.value), n_blocks=int(nb.value),
                    task=task[:nt], span=span[:nsp], reach=reach[:nr], fetch=fetch[:nf], blk_coff=blk_coff[:nblk], blk_crc=blk_crc[:nblk], n_ref=len(self.contigs))
        t1 = time.perf_counter()
        if join is not None:
            return self._joined_on_the_device(join, sh, plan, n_host, nt, int(min_base_qual), bool(all_bases), release, t0, t1)
        if walk is None:  # the host's twin
            d_first, d_walked = np.zeros(n_host + 1, np.int64), np.zeros(max(1, n_host), np.int64)
            _check(self.lib, self.lib.uz_stage_walk_host(sh.ptr, None, 0, d_first.ctypes.data, d_walked.ctypes.data))
            desc = np.zeros(max(1, int(d_first[-1])), WALK_DESC)
            _check(self.lib, self.lib.uz_stage_walk_host(sh.ptr, desc.ctypes.data, int(desc.size), d_first.ctypes.data, d_walked.ctypes.data))
            desc = desc[: int(d_first[-1])]
            d_flags, token = np.zeros(max(1, n_host), np.int32), None
        else:
            desc, d_first, d_flags, d_walked, token = walk(plan)
            d_first = np.ascontiguousarray(d_first, np.int64)
            d_flags = np.ascontiguousarray(d_flags, np.int32)
            d_walked = np.ascontiguousarray(d_walked, np.int64)
            assert d_first.size == nt + 1
            if merge:  # the device walked the plan's tasks: joined per task of the stage (a record two neighbouring sub-tasks met is kept once)
                h_first, h_flags, h_walked = np.zeros(n_host + 1, np.int64), np.zeros(max(1, n_host), np.int32), np.zeros(max(1, n_host), np.int64)
                desc = np.ascontiguousarray(desc)
                try:
                    _check(self.lib, self.lib.uz_stage_merge_subtasks(sh.ptr, desc.ctypes.data, d_first.ctypes.data, d_flags.ctypes.data, d_walked.ctypes.data,
                                                                      h_first.ctypes.data, h_flags.ctypes.data, h_walked.ctypes.data))
                except BaseException:
                    if release is not None and token is not None:
                        release(token)
                    raise
                desc, d_first, d_flags, d_walked = desc[: int(h_first[-1])], h_first, h_flags[:n_host], h_walked[:n_host]
        by_sub = walk is not None and not merge  # (the joins read the sub-tasks' descriptors as they came: uz_bam_stage_finish_sub)
        nt = nt if by_sub else n_host
        t2 = time.perf_counter()
        out = KeptBatch()
        out.token, out._release = token, release
        d_first = np.ascontiguousarray(d_first, np.int64)
        d_flags = np.ascontiguousarray(d_flags, np.int32)
        d_walked = np.ascontiguousarray(d_walked, np.int64)
        assert desc.dtype == WALK_DESC and d_first.size == nt + 1
        _check(self.lib, (self.lib.uz_bam_stage_finish_sub if by_sub else self.lib.uz_bam_stage_finish_desc)(
            sh.ptr, desc.ctypes.data, d_first.ctypes.data, d_flags.ctypes.data, d_walked.ctypes.data))
        t3 = time.perf_counter()
        self.lib.uz_stage_kept_sizes(sh.ptr, z)
        out.n, out.n_cigar_total, out.n_row_units, out.n_seq_units, out.n_qnames, n_aux, out.host_tasks, out.n_name_bytes = (int(x) for x in z[:8])
        out.kept = (alloc(max(1, out.n) * KEPT_REC.itemsize).view(KEPT_REC) if alloc else np.zeros(max(1, out.n), KEPT_REC))[: out.n]
        out.contig_off = np.zeros(len(self.contigs) + 1, np.int64)
        out.max_span = np.zeros(max(1, len(self.contigs)), np.int32)
        out.aux = ia(n_aux + 64)
        _check(self.lib, self.lib.uz_stage_kept(sh.ptr, int(self.threads), out.kept.ctypes.data, out.contig_off.ctypes.data, out.max_span.ctypes.data,
                                                out.aux.ctypes.data, int(n_aux)))
        out.n_aux = n_aux
        t4 = time.perf_counter()
        io = (C.c_int64 * 8)()
        self.lib.uz_stage_io_stats(sh.ptr, io)
        out.io_stats = dict(zip(("file_bytes_read", "blocks_inflated", "records_walked", "records_kept", "reach_intervals", "index_mate_lookups",
                                 "blocks_from_the_device", "gathered_bytes"), (int(x) for x in io)))
        tm = (C.c_double * 6)()
        self.lib.uz_stage_timing(sh.ptr, tm)
        out.timing = dict(plan=t1 - t0, walk=t2 - t1, joins=t3 - t2, kept=t4 - t3, mates=float(tm[2]), numbering=float(tm[3]))
        out.plan, out.desc, out.d_first, out.d_flags = plan, desc, d_first, d_flags
        out.min_base_qual, out.n_contigs, out.all_bases = int(min_base_qual), len(self.contigs), bool(all_bases)
        out._stage = sh
        return out
