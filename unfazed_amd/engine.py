"""ctypes binding of libunfazed_hip.so (include/unfazed_hip.h) and the backend
object the host path drives.  There is no CPU fallback: if the library or a
gfx950 device is missing, construction raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UZ_HIP_LIB", os.path.join(_HERE, "libunfazed_hip.so"))

K_SITE_SCAN, K_WINDOW_COUNT, K_WINDOW_FILL, K_PHASE, K_SIZING, K_CNV = 0, 1, 2, 3, 4, 5

EXPORTS = [
    "uz_create", "uz_destroy", "uz_last_error", "uz_sync", "uz_set_params",
    "uz_sites_upload", "uz_family_upload", "uz_sites_family_upload_async", "uz_reads_upload", "uz_reads_upload_packed", "uz_reads_wait", "uz_reads_headers", "uz_bgzf_inflate", "uz_bgzf_inflate_to_host", "uz_bam_walk", "uz_crc32_blocks", "uz_bam_walk_fetch", "uz_bam_walk_release", "uz_reads_from_bam",
    "uz_bam_walk_flags", "uz_bam_join", "uz_bam_join_needs", "uz_bam_join_fetch", "uz_reads_from_walk", "uz_reads_names", "uz_walk_slot_stats", "uz_walk_reserve",
    "uz_pinned_alloc", "uz_pinned_free",
    "uz_sites_adopt_device", "uz_family_adopt_device", "uz_reads_adopt_device",
    "uz_sites_free", "uz_reads_free", "uz_drop_derived",
    "uz_site_scan", "uz_site_scan_many", "uz_site_classes", "uz_find", "uz_find_fetch",
    "uz_phase", "uz_phase_begin", "uz_phase_end", "uz_phase_cohort", "uz_phase_votes", "uz_phase_groups", "uz_phase_cnv", "uz_phase_cnv_sites",
    "uz_prof_enable", "uz_prof_reset", "uz_prof_get", "uz_prof_units",
]

_lib = None


class UnfazedHipError(RuntimeError):
    pass


def load_library(path: Optional[str] = None):
    """dlopen the HIP library and declare the prototypes.  Fails loudly."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise UnfazedHipError(
            "%s not found: build it with `python -m unfazed_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the phasing path." % p
        )
    L = C.CDLL(p)
    vp, i32, i64p = C.c_void_p, C.c_int, C.c_void_p
    L.uz_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.uz_destroy.argtypes = [vp]
    L.uz_destroy.restype = None
    L.uz_last_error.argtypes = [vp]
    L.uz_last_error.restype = C.c_char_p
    L.uz_sync.argtypes = [vp]
    L.uz_set_params.argtypes = [vp, vp]
    L.uz_reads_wait.argtypes = [vp, C.c_int]
    L.uz_reads_headers.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.uz_bgzf_inflate.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, vp, C.c_int, C.POINTER(C.c_double)]
    L.uz_bgzf_inflate_to_host.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, vp]
    L.uz_crc32_blocks.argtypes = [vp, vp, C.c_int64, vp, vp, vp]
    L.uz_bam_walk.argtypes = [vp, vp, C.c_int64, C.c_int64, vp, vp, vp, vp, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, vp, vp]
    L.uz_bam_walk_fetch.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.uz_bam_walk_release.argtypes = [vp, C.c_int]
    L.uz_reads_from_bam.argtypes = [vp, C.c_int, vp, C.c_int64, vp, C.c_int64, vp, vp, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_uint32, C.c_int32, vp, C.c_int64, vp]
    L.uz_bam_walk_flags.argtypes = [vp, C.c_int, vp, vp]
    L.uz_bam_join.argtypes = [vp, C.c_int, C.c_int32, vp, C.c_int32, C.c_int, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, vp, vp]
    L.uz_bam_join_needs.argtypes = [vp, C.c_int, vp]
    L.uz_bam_join_fetch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.uz_reads_from_walk.argtypes = [vp, C.c_int, C.c_int32, C.c_int, vp, vp]
    L.uz_reads_names.argtypes = [vp, C.c_int, vp, C.c_int64, vp, vp]
    L.uz_walk_slot_stats.argtypes = [vp, vp]
    L.uz_walk_reserve.argtypes = [vp, C.c_int]
    L.uz_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.uz_pinned_free.argtypes = [vp]
    L.uz_pinned_free.restype = None
    for f in ("uz_sites_upload", "uz_reads_upload", "uz_reads_upload_packed", "uz_sites_adopt_device", "uz_reads_adopt_device"):
        getattr(L, f).argtypes = [vp, vp, C.POINTER(C.c_int)]
    for f in ("uz_family_upload", "uz_family_adopt_device"):
        getattr(L, f).argtypes = [vp, C.c_int, vp, C.POINTER(C.c_int)]
    L.uz_sites_family_upload_async.argtypes = [vp, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.uz_drop_derived.argtypes = [vp]
    L.uz_sites_free.argtypes = [vp, C.c_int]
    L.uz_reads_free.argtypes = [vp, C.c_int]
    L.uz_site_scan.argtypes = [vp, C.c_int]
    L.uz_site_scan_many.argtypes = [vp, vp, C.c_int32]
    L.uz_site_classes.argtypes = [vp, C.c_int, vp]
    L.uz_find.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
    L.uz_find_fetch.argtypes = [vp, vp, vp, vp]
    L.uz_phase.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    L.uz_phase_begin.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
    L.uz_phase_end.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    L.uz_phase_cohort.argtypes = [vp, vp, C.c_int32, vp, C.c_int, vp, vp, vp, vp]
    L.uz_phase_cnv.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.uz_phase_cnv_sites.argtypes = [vp, vp, vp]
    L.uz_phase_votes.argtypes = [vp, vp, vp]
    L.uz_phase_groups.argtypes = [vp, vp, vp]
    L.uz_prof_enable.argtypes = [vp, C.c_int]
    L.uz_prof_reset.argtypes = [vp]
    L.uz_prof_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.uz_prof_units.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("uz_destroy", "uz_last_error", "uz_pinned_free"):
            fn.restype = C.c_int
    if path is None:
        _lib = L
    return L


class _CappedWalk:
    """the engine's walk / join primitives for io_native.BamSource.select_kept(join=...), with the cap on a walked batch's inflated bytes applied
    where the plan is first seen"""

    def __init__(self, eng, cap: int):
        self._eng, self._cap = eng, int(cap)

    def walk(self, plan):
        if int(plan["out_bytes"]) > self._cap:
            raise _WalkTooLarge()
        return self._eng.walk(plan)

    def __getattr__(self, name):
        return getattr(self._eng, name)


class _WalkTooLarge(Exception):
    """the blocks of a batch inflate to more than the device walk keeps in HBM (HipEngine.stage_reads falls back to the host route)"""


class HipEngine:
    """One context on one GPU.  Backend interface used by hostpath.PhasingHost."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.uz_create(int(device), C.byref(h))
        if rc != 0:
            raise UnfazedHipError(
                "uz_create(device=%d) failed with %d: a gfx950 (MI355X) device is required; "
                "there is no CPU fallback" % (device, rc)
            )
        self.h = h
        self.device = device
        self._keep: List[object] = []
        self._staged = {}
        self._staged_sites = {}
        self._params = None
        import threading
        self._inflate_lock = threading.Lock()
        # walked batches on the device at once, each on its slot's own streams.  Round 5 (the host's joins the bound): two measured 10 % slower than
        # one; with the joins on the device the device's chain is the bound and a batch's walk -- a latency-bound kernel of ~1 900 wavefronts -- leaves
        # the chip to the next batch's inflate: 1 / 2 / 3 at once = 117 / 130 / 128 k DNMs/s in bench.py's feed pass
        self._walk_sem = threading.BoundedSemaphore(int(os.environ.get("UZ_WALKS_AT_ONCE", "2")))
        self._inflate_bufs = None  # PinnedPair of upload_reads_staged
        self._stage_pool = None    # PinnedPool of upload_reads_staged: the staged columns' page-locked block, kept from batch to batch
        self._chunk_pools = [None, None]  # stage_reads: two alternating sets of page-locked buffers (columns; gathered / inflated blocks)
        self._chunk_pairs = [None, None]

    def close(self):
        if getattr(self, "_inflate_bufs", None) is not None:
            self._inflate_bufs.free_all()
            self._inflate_bufs = None
        if getattr(self, "_stage_pool", None) is not None:
            self._stage_pool.free_all()
            self._stage_pool = None
        for k in range(len(getattr(self, "_chunk_pools", []))):
            if self._chunk_pools[k] is not None:
                self._chunk_pools[k].free_all()
                self._chunk_pairs[k].free_all()
                self._chunk_pools[k] = self._chunk_pairs[k] = None
        if getattr(self, "h", None):
            self.L.uz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            msg = self.L.uz_last_error(self.h)
            raise UnfazedHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    # -------------------------------------------------------------- staging
    def set_params(self, params: abi.Params):
        self._ck(self.L.uz_set_params(self.h, C.byref(params)), "uz_set_params")
        self._params = params

    def upload_sites(self, sites) -> int:
        v = abi.sites_view(sites)
        sid = C.c_int(-1)
        self._ck(self.L.uz_sites_upload(self.h, v.ref(), C.byref(sid)), "uz_sites_upload")
        return sid.value

    def upload_sites_view(self, held: abi.Held) -> int:
        """the same from a prepared view (e.g. columns in pinned memory)"""
        sid = C.c_int(-1)
        self._ck(self.L.uz_sites_upload(self.h, held.ref(), C.byref(sid)), "uz_sites_upload")
        return sid.value

    def add_family(self, sites_h: int, gt, rd, ad, gq, wide=None) -> int:
        v = abi.family_view(gt, rd, ad, gq, wide)
        fid = C.c_int(-1)
        self._ck(self.L.uz_family_upload(self.h, int(sites_h), v.ref(), C.byref(fid)), "uz_family_upload")
        return fid.value

    def upload_sites_family_async(self, held: abi.Held, gt, rd, ad, gq, wide=None):
        """sites + one trio's genotype columns queued on the copy stream (uz_sites_family_upload_async) -> (sites id, family id);
        the arrays (pinned) must stay alive until a call using the family has returned: the engine keeps them.
        rd / ad / gq as uint8 arrays = the eight-bit link form (abi.family_columns8, which also extends `wide`)."""
        v = abi.family_view(gt, rd, ad, gq, wide)
        sid, fid = C.c_int(-1), C.c_int(-1)
        self._ck(self.L.uz_sites_family_upload_async(self.h, held.ref(), v.ref(), C.byref(sid), C.byref(fid)), "uz_sites_family_upload_async")
        self._staged_sites[sid.value] = (held, v)
        return sid.value, fid.value

    def upload_reads(self, reads, min_base_qual=None, point_only=False, fetches=None, all_bases=False, wide_no_units=False) -> int:
        """A decoded table -> HBM.  With the base-quality threshold of the run (min_base_qual = --min-gt-qual) the table
        goes over the link in the staged form (packed on the host into pinned memory, several times fewer bytes); without it
        in the ASCII form, which the device packs and which then serves any threshold.
        point_only: the table will only serve batches of point variants (SNV / indel) -- the qualities then travel as
        per-record counts + short position lists instead of the one-bit plane (uz_types.h: the list form is exact for "good"
        records, and a point-variant batch never looks at the bits of any other; an SV batch does).
        fetches = (contig, lo, hi, extra) of staging.fetch_points for the ONE batch the table will serve (point_only): records no
        fetch returns travel without bases, of the others' rows only the 32-base units that hold a fetched position (all_bases:
        --no-extended batches read mates at candidate sites nobody fetched -- every record keeps its rows).
        wide_no_units: the batch holds SVs -- their +-cutoff fetches stage no base unit (collect_reads_sv reads none); the list form of
        the qualities serves them too: the read stage asks for quality bits of records that pass goodread only (uz_types.h)."""
        v = abi.reads_view(reads)
        if min_base_qual is not None:
            from . import io_native
            pool = PinnedPool()
            rid = None
            try:
                # the staged columns back to back in one pinned block (they cross the link as one copy); the packed form is at most
                # half the size of the ASCII table, usually a small fraction
                pool.new_slab(sum(int(getattr(x, "nbytes", 0)) for x in v.arrays.values()) // 2 + (1 << 20))
                if fetches is not None and point_only:
                    full = io_native.pack_reads(v, int(min_base_qual), lists=True, with_end=True)
                    fc, flo, fhi, fex = fetches
                    packed, idx = io_native.ReadsSource(full).select(fc, flo, fhi, alloc=pool.alloc, want_index=True, all_bases=bool(all_bases),
                                                                     extra=fex, wide_no_units=bool(wide_no_units))
                    if idx.size != int(v.view.n_segs):  # record numbers are the caller's: the table must be the fetches' own reach
                        raise UnfazedHipError("upload_reads(fetches=...): the table holds records the fetches cannot reach")
                    if packed.qname_map is not None and not np.array_equal(packed.qname_map, np.arange(packed.qname_map.size, dtype=np.uint32)):
                        # name ids are the caller's too: the pair form renumbers them unless they already count first appearances
                        pool.rewind(0)
                        packed, idx = io_native.ReadsSource(full).select(fc, flo, fhi, alloc=pool.alloc, want_index=True, all_bases=bool(all_bases),
                                                                         extra=fex, wide_no_units=bool(wide_no_units), pair8=False)
                else:
                    packed = io_native.pack_reads(v, int(min_base_qual), alloc=pool.alloc, lists=bool(point_only), with_end=None, cigar_compact=True)
                rid = self.upload_reads_packed(packed)
                self.wait_reads(rid)  # the pinned buffers go back right away: the caller may drop the table
                self._staged.pop(rid, None)
            except BaseException:
                if rid is not None:  # the copy may still be reading the pinned block: let it finish before the block goes back
                    try:
                        self.wait_reads(rid)
                    except UnfazedHipError:
                        pass
                    self._staged.pop(rid, None)
                raise
            finally:
                pool.free_all()
            return rid
        rid = C.c_int(-1)
        self._ck(self.L.uz_reads_upload(self.h, v.ref(), C.byref(rid)), "uz_reads_upload")
        return rid.value

    def upload_reads_staged(self, src, fc, flo, fhi, fex, min_base_qual: int, all_bases: bool = False, wide_no_units: bool = False):
        """One batch straight from an indexed BAM (io_native.BamSource): the records its fetches return + their mates, built in the
        link form in pinned memory by one pass over the file's blocks (uz_bam_stage_*), uploaded as one table.
        -> (reads id, the staged view: `.qnames` maps the name ids of the result lists back to strings)"""
        # the page-locked block of the staged columns is kept from batch to batch (pinning a gigabyte takes about a second: round 3's
        # product route spent 1.2 s of its 2.4 s per 20 k DNMs there): rewound when it is large enough, replaced when not
        if self._stage_pool is None:
            self._stage_pool = PinnedPool()
        pool = self._stage_pool
        pool.keep = True
        rid = None
        if os.environ.get("UZ_INFLATE", "device") == "device" and os.environ.get("UZ_WALK", "device") == "device":
            # the record walk on the device (include/uz_bamwalk.h; stage_reads has the chunked form): blocks inflated, checked and walked in HBM, the
            # joins here on descriptors, the table unpacked where the records lie.  UZ_WALK=host: the link form below
            if self._inflate_bufs is None:
                self._inflate_bufs = PinnedPair()
            pair = self._inflate_bufs
            pair.start()
            cap = int(os.environ.get("UZ_WALK_MAX_BYTES", 16 << 30))  # (stage_reads: a batch beyond it takes the host route below)

            def walk(plan):
                if int(plan["out_bytes"]) > cap:
                    raise _WalkTooLarge()
                return self.bam_walk(plan, alloc=pair.alloc)
            try:
                # (UZ_JOINS=host: the descriptors come down and the host joins them, round 5's route; default: the joins on the device too, k_bamjoin.hip)
                dev_joins = os.environ.get("UZ_JOINS", "device") == "device"
                kb = src.select_kept(fc, flo, fhi, int(min_base_qual), walk=None if dev_joins else walk, join=_CappedWalk(self, cap) if dev_joins else None,
                                     all_bases=bool(all_bases), alloc=pair.alloc, extra=fex, release=self.bam_walk_release)
            except _WalkTooLarge:
                kb = None
            if kb is not None:
                rid = self.reads_from_bam(kb, names=True)
                if getattr(kb, "joined", False):  # (this caller may free the table before it asks for names: they come down now, all of them)
                    kb.qnames = kb.qnames.frozen()
                names = type("StagedNames", (), {})()
                names.qnames, names.io_stats, names.timing = kb.qnames, kb.io_stats, kb.timing
                return rid, names
        try:
            inflate = inflate_alloc = None
            if os.environ.get("UZ_INFLATE", "device") == "device":  # the batch's BGZF blocks inflated on the device (UZ_INFLATE=host: by the host's cores)
                if self._inflate_bufs is None:
                    self._inflate_bufs = PinnedPair()
                self._inflate_bufs.start()
                inflate, inflate_alloc = self.inflate_blocks, self._inflate_bufs.alloc
            packed = src.select(fc, flo, fhi, int(min_base_qual), pool=pool, all_bases=bool(all_bases), extra=fex, wide_no_units=bool(wide_no_units),
                                inflate=inflate, inflate_alloc=inflate_alloc)
            pool.end_slab()
            rid = self.upload_reads_packed(packed)
            self.wait_reads(rid)  # the pinned buffers go back right away
            self._staged.pop(rid, None)
        except BaseException:
            if rid is not None:  # the copy may still be reading the pinned block: let it finish before the block goes back
                try:
                    self.wait_reads(rid)
                except UnfazedHipError:
                    pass
                self._staged.pop(rid, None)
            raise
        names = type("StagedNames", (), {})()
        names.qnames, names.io_stats, names.timing = packed.qnames, packed.io_stats, packed.timing
        return rid, names

    def stage_reads(self, src, fc, flo, fhi, fex, min_base_qual: int, all_bases: bool = False, wide_no_units: bool = False, slot: int = 0):
        """The first half of upload_reads_staged alone -- the batch's records built in the link form in page-locked memory -- for a caller that
        overlaps it with the device work of the batch before (hostpath: chunks of a large batch; may be called from a worker thread, the
        device inflates the BGZF blocks on streams of its own).  slot: which set of page-locked buffers to use (sets are made as they are asked
        for); the block of a slot is re-used by the next stage_reads on it, so the table staged there must have landed on the device by then.
        -> the packed view for upload_reads_packed (`.qnames`, `.io_stats`, `.timing` ride on it)"""
        slot = int(slot)
        while len(self._chunk_pools) <= slot:
            self._chunk_pools.append(None)
            self._chunk_pairs.append(None)
        if self._chunk_pools[slot] is None:
            self._chunk_pools[slot] = PinnedPool()
            self._chunk_pools[slot].keep = True
            self._chunk_pairs[slot] = PinnedPair()
        pool, pair = self._chunk_pools[slot], self._chunk_pairs[slot]
        inflate = inflate_alloc = None
        if os.environ.get("UZ_INFLATE", "device") == "device" and os.environ.get("UZ_WALK", "device") == "device":
            # the record walk on the device too (include/uz_bamwalk.h): the batch's blocks are inflated AND walked in HBM, the host runs the joins on
            # 64-byte descriptors, and upload_reads_packed builds the table from the bytes the walk left on the device (UZ_WALK=host: the link form)
            from . import io_native
            pair.start()
            # a walked batch keeps its compressed blocks, the inflated bytes and worst-case descriptor slices (64 B per 36 B of inflated data) in
            # HBM until its table is built, in one of four slots: a batch whose blocks inflate to more than the cap takes the host route below
            # (the link form: nothing but the packed table reaches the device) instead of failing in hipMalloc
            cap = int(os.environ.get("UZ_WALK_MAX_BYTES", 16 << 30))

            def walk(plan):
                if int(plan["out_bytes"]) > cap:
                    raise _WalkTooLarge()
                return self.bam_walk(plan, alloc=pair.alloc)
            try:
                dev_joins = os.environ.get("UZ_JOINS", "device") == "device"  # (as upload_reads_staged)
                kb = src.select_kept(fc, flo, fhi, int(min_base_qual), walk=None if dev_joins else walk, join=_CappedWalk(self, cap) if dev_joins else None,
                                     all_bases=bool(all_bases), alloc=pair.alloc, extra=fex, release=self.bam_walk_release)
                kb._alloc = pair.alloc  # (host joins: the names of its records come back into page-locked memory of the same set: reads_from_bam)
                return kb
            except _WalkTooLarge:
                pass
        if os.environ.get("UZ_INFLATE", "device") == "device":
            pair.start()
            inflate, inflate_alloc = self.inflate_blocks, pair.alloc
        packed = src.select(fc, flo, fhi, int(min_base_qual), pool=pool, all_bases=bool(all_bases), extra=fex, wide_no_units=bool(wide_no_units),
                            inflate=inflate, inflate_alloc=inflate_alloc)
        pool.end_slab()
        return packed

    def upload_reads_packed(self, packed: abi.Held) -> int:
        """Staged form, asynchronous: the arrays of `packed` must stay alive and untouched until wait_reads() or
        a phase on the table has returned (they are kept referenced here until the table is freed)."""
        if getattr(packed, "token", None) is not None and hasattr(packed, "kept"):  # a batch walked on the device (stage_reads): its table is built from HBM
            return self.reads_from_bam(packed, names=True)
        rid = C.c_int(-1)
        self._ck(self.L.uz_reads_upload_packed(self.h, packed.ref(), C.byref(rid)), "uz_reads_upload_packed")
        self._staged[rid.value] = packed
        return rid.value

    def wait_reads(self, rid: int):
        self._ck(self.L.uz_reads_wait(self.h, int(rid)), "uz_reads_wait")

    def reads_headers(self, rid: int, n: int) -> dict:
        """start / end / tlen / mate / qname of a table as the device holds them (uz_reads_headers)"""
        out = {k: np.zeros(max(1, int(n)), np.uint32 if k == "qname" else np.int32) for k in ("start", "end", "tlen", "mate", "qname")}
        self._ck(self.L.uz_reads_headers(self.h, int(rid), *(out[k].ctypes.data for k in ("start", "end", "tlen", "mate", "qname"))), "uz_reads_headers")
        return {k: v[: int(n)] for k, v in out.items()}

    def bgzf_inflate(self, data, repeat: int = 0):
        """Every BGZF block of `data` (bytes / uint8 array: a BGZF file or a run of its blocks) inflated on the device (uz_bgzf_inflate).
        -> (uint8 array of the inflated bytes, block count, mean kernel ms over `repeat` extra runs or None)"""
        buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, np.uint8)
        n, at, ins, sizes = buf.size, 0, [], []
        while at + 18 <= n:
            if not (buf[at] == 0x1F and buf[at + 1] == 0x8B and buf[at + 2] == 8 and (buf[at + 3] & 4)):
                raise UnfazedHipError("not a BGZF block at byte %d" % at)
            xlen = int(buf[at + 10]) | (int(buf[at + 11]) << 8)
            bsize, p = None, at + 12
            while p + 4 <= at + 12 + xlen:
                slen = int(buf[p + 2]) | (int(buf[p + 3]) << 8)
                if buf[p] == 66 and buf[p + 1] == 67 and slen == 2:
                    bsize = (int(buf[p + 4]) | (int(buf[p + 5]) << 8)) + 1
                p += 4 + slen
            if bsize is None or at + bsize > n:
                raise UnfazedHipError("BGZF block at byte %d: no BC field, or the block overruns the data" % at)
            ins.append(at + 12 + xlen)
            sizes.append(int(buf[at + bsize - 4]) | (int(buf[at + bsize - 3]) << 8) | (int(buf[at + bsize - 2]) << 16) | (int(buf[at + bsize - 1]) << 24))
            at += bsize
        nb = len(ins)
        in_off = np.asarray(ins, np.int64)
        out_off = np.concatenate([[0], np.cumsum(np.asarray(sizes, np.int64))]).astype(np.int64)
        out = np.zeros(max(1, int(out_off[-1])), np.uint8)
        ms = C.c_double(0.0)
        self._ck(self.L.uz_bgzf_inflate(self.h, buf.ctypes.data, int(n), nb, in_off.ctypes.data if nb else None, out_off.ctypes.data, out.ctypes.data,
                                        int(repeat), C.byref(ms)) if nb else 0, "uz_bgzf_inflate")
        return out[: int(out_off[-1])], nb, (ms.value if repeat > 0 else None)

    def inflate_blocks(self, comp: np.ndarray, comp_bytes: int, in_off: np.ndarray, out_off: np.ndarray, out: np.ndarray):
        """The gathered BGZF blocks of a staged batch (io_native.BamSource.select(inflate=engine.inflate_blocks)) inflated on the device:
        comp[:comp_bytes] -> out[:out_off[-1]] (both best in pinned memory: PinnedPair).  May be called from a decoder's worker thread
        beside the main thread's calls (its own stream and buffers; one call at a time)."""
        with self._inflate_lock:
            rc = self.L.uz_bgzf_inflate_to_host(self.h, comp.ctypes.data, int(comp_bytes), int(in_off.size), in_off.ctypes.data, out_off.ctypes.data,
                                                out.ctypes.data)
            if rc != 0:
                raise UnfazedHipError("uz_bgzf_inflate_to_host: %s" % (self.L.uz_last_error(self.h) or b"").decode(errors="replace"))

    # ---- the record walk on the device (include/uz_bamwalk.h)
    def bam_walk(self, plan: dict, alloc=None):
        """io_native.BamSource.select_kept(walk=engine.bam_walk): the batch's gathered BGZF blocks inflated in HBM and walked there
        (uz_bam_walk + uz_bam_walk_fetch).  -> (descriptors, d_first, d_flags, d_walked, walk id); the inflated bytes wait on the device for
        reads_from_bam (or bam_walk_release).  May be called from a decoder's worker thread."""
        from . import io_native
        wid, nd = C.c_int(-1), C.c_int64(0)
        nt = int(plan["task"].shape[0])
        with self._walk_sem:  # (UZ_WALKS_AT_ONCE batches at a time)
            rc = self.L.uz_bam_walk(self.h, plan["comp"].ctypes.data, int(plan["comp_bytes"]), int(plan["n_blocks"]), plan["in_off"].ctypes.data,
                                    plan["out_off"].ctypes.data, plan["blk_coff"].ctypes.data,
                                    plan["blk_crc"].ctypes.data if plan.get("blk_crc") is not None and os.environ.get("UZ_WALK_CRC", "1") != "0" else None, nt, plan["task"].ctypes.data, int(plan["span"].shape[0]),
                                    plan["span"].ctypes.data, int(plan["reach"].shape[0]), plan["reach"].ctypes.data, int(plan["fetch"].shape[0]),
                                    plan["fetch"].ctypes.data, C.byref(wid), C.byref(nd))
            if rc != 0:
                raise UnfazedHipError("uz_bam_walk: %s" % (self.L.uz_last_error(self.h) or b"").decode(errors="replace"))
            n = int(nd.value)
            desc = (alloc(max(1, n) * 64).view(io_native.WALK_DESC) if alloc else np.zeros(max(1, n), io_native.WALK_DESC))[: max(1, n)]
            d_first, d_flags, d_walked = np.zeros(nt + 1, np.int64), np.zeros(max(1, nt), np.int32), np.zeros(max(1, nt), np.int64)
            rc = self.L.uz_bam_walk_fetch(self.h, wid.value, desc.ctypes.data, d_first.ctypes.data, d_flags.ctypes.data, d_walked.ctypes.data)
            if rc != 0:
                self.L.uz_bam_walk_release(self.h, wid.value)
                raise UnfazedHipError("uz_bam_walk_fetch: %s" % (self.L.uz_last_error(self.h) or b"").decode(errors="replace"))
        return desc[:n], d_first, d_flags[:nt], d_walked[:nt], wid.value

    # ---- the joins of a walked batch on the device (csrc/k_bamjoin.hip): the primitives io_native.BamSource.select_kept(join=engine) drives
    def walk(self, plan: dict):
        """uz_bam_walk alone: blocks up, inflated, checked and walked in HBM -> (walk id, descriptors the joins can need).  Nothing comes down."""
        wid, nd = C.c_int(-1), C.c_int64(0)
        nt = int(plan["task"].shape[0])
        with self._walk_sem:
            rc = self.L.uz_bam_walk(self.h, plan["comp"].ctypes.data, int(plan["comp_bytes"]), int(plan["n_blocks"]), plan["in_off"].ctypes.data,
                                    plan["out_off"].ctypes.data, plan["blk_coff"].ctypes.data,
                                    plan["blk_crc"].ctypes.data if plan.get("blk_crc") is not None and os.environ.get("UZ_WALK_CRC", "1") != "0" else None, nt, plan["task"].ctypes.data, int(plan["span"].shape[0]),
                                    plan["span"].ctypes.data, int(plan["reach"].shape[0]), plan["reach"].ctypes.data, int(plan["fetch"].shape[0]),
                                    plan["fetch"].ctypes.data, C.byref(wid), C.byref(nd))
            if rc != 0:
                raise UnfazedHipError("uz_bam_walk: %s" % (self.L.uz_last_error(self.h) or b"").decode(errors="replace"))
        return wid.value, int(nd.value)

    def walk_flags(self, token: int, nt: int):
        d_flags, d_walked = np.zeros(max(1, nt), np.int32), np.zeros(max(1, nt), np.int64)
        self._ck(self.L.uz_bam_walk_flags(self.h, int(token), d_flags.ctypes.data, d_walked.ctypes.data), "uz_bam_walk_flags")
        return d_flags[:nt], d_walked[:nt]

    def join(self, token: int, n_host: int, h_flags, n_ref: int, all_bases: bool, xdesc=None, xaux=None, look_tid=None, need_jtask=None):
        """one call of uz_bam_join -> (needs, totals [8])"""
        nn, tot = C.c_int64(0), np.zeros(8, np.int64)
        nx = 0 if xdesc is None else int(xdesc.size)
        na = 0 if xaux is None else int(xaux.size)
        nl = 0 if look_tid is None else int(look_tid.size)
        self._ck(self.L.uz_bam_join(self.h, int(token), int(n_host), h_flags.ctypes.data if h_flags is not None else None, int(n_ref), 1 if all_bases else 0,
                                      xdesc.ctypes.data if nx else None, nx, xaux.ctypes.data if na else None, na, look_tid.ctypes.data if nl else None, nl,
                                      need_jtask.ctypes.data if need_jtask is not None else None, C.byref(nn), tot.ctypes.data), "uz_bam_join")
        return int(nn.value), tot

    def join_needs(self, token: int, n: int):
        from . import io_native
        need = np.zeros(max(1, n), io_native.NEED_REC)
        self._ck(self.L.uz_bam_join_needs(self.h, int(token), need.ctypes.data), "uz_bam_join_needs")
        return need[:n]

    def join_fetch(self, token: int, n: int, n_ref: int):
        """parity / debug: the kept records of a finished join in output order -> dict(voff, qname, mate, bases, kept, contig_off, max_span)"""
        from . import io_native
        out = dict(voff=np.zeros(max(1, n), np.uint64), qname=np.zeros(max(1, n), np.uint32), mate=np.zeros(max(1, n), np.int32), bases=np.zeros(max(1, n), np.uint8),
                   kept=np.zeros(max(1, n), io_native.KEPT_REC), contig_off=np.zeros(n_ref + 1, np.int64), max_span=np.zeros(max(1, n_ref), np.int32))
        self._ck(self.L.uz_bam_join_fetch(self.h, int(token), *(out[k].ctypes.data for k in ("voff", "qname", "mate", "bases", "kept", "contig_off", "max_span"))), "uz_bam_join_fetch")
        return {k: (v[:n] if k in ("voff", "qname", "mate", "bases", "kept") else v) for k, v in out.items()}

    def reads_names_raw(self, rid: int, ids):
        """the read names of name ids of a table built from a batch joined on the device (uz_reads_names) -> (bytes back to back, off [n + 1])"""
        ids = np.ascontiguousarray(ids, np.uint32)
        n = int(ids.size)
        off, ptr = np.zeros(n + 1, np.int64), C.c_void_p()
        if n == 0:
            return np.zeros(1, np.uint8), off
        self._ck(self.L.uz_reads_names(self.h, int(rid), ids.ctypes.data, n, off.ctypes.data, C.byref(ptr)), "uz_reads_names")
        total = int(off[-1])
        buf = np.frombuffer(C.string_at(ptr.value, total), np.uint8) if total else np.zeros(1, np.uint8)  # (a copy: the context's block is the next call's)
        return buf, off

    def reads_names(self, rid: int, ids) -> list:
        from .io_native import names_of_buffer
        return names_of_buffer(*self.reads_names_raw(rid, ids))

    def walk_slot_stats(self) -> dict:
        z = np.zeros(8, np.int64)
        self._ck(self.L.uz_walk_slot_stats(self.h, z.ctypes.data), "uz_walk_slot_stats")
        return dict(allocations=int(z[0]), parked_blocks=int(z[1]), parked_bytes=int(z[2]), slot_room=[int(x) for x in z[4:8]])

    def crc32_blocks(self, data: np.ndarray, off: np.ndarray, want: np.ndarray) -> int:
        """k_bgzf_crc32 on blocks data[off[k]:off[k+1]] (<= 64 KiB each) against want[k] -> -1, or a block whose CRC-32 differs (uz_crc32_blocks)"""
        data = np.ascontiguousarray(data, np.uint8)
        off = np.ascontiguousarray(off, np.int64)
        want = np.ascontiguousarray(want, np.uint32)
        bad = C.c_int64(-1)
        self._ck(self.L.uz_crc32_blocks(self.h, data.ctypes.data, int(off.size) - 1, off.ctypes.data, want.ctypes.data, C.byref(bad)), "uz_crc32_blocks")
        return int(bad.value)

    def bam_walk_release(self, walk_id: int):
        self._ck(self.L.uz_bam_walk_release(self.h, int(walk_id)), "uz_bam_walk_release")

    def reads_from_bam(self, kb, names: bool = False) -> int:
        """The table of a batch walked on the device (kb: io_native.KeptBatch of select_kept(walk=self.bam_walk)): unpacked from the bytes in HBM.
        names: the read names of the kept records come back too (kb.qnames then maps the name ids of the result lists to strings)."""
        rid = C.c_int(-1)
        if getattr(kb, "joined", False):  # the joins ran on the device: the kept list lies in the batch's slot (uz_reads_from_walk)
            from . import io_native
            tot = np.zeros(8, np.int64)
            self._ck(self.L.uz_reads_from_walk(self.h, int(kb.token), int(kb.min_base_qual), 1 if names else 0, C.byref(rid), tot.ctypes.data), "uz_reads_from_walk")
            kb.token = None
            # a batch is through and its sizes are known: the other slots a pipeline will use grow to them now, not inside a later batch's walk
            # (a no-op once they have; UZ_WALK_RESERVE=0 leaves the slots to grow at first use)
            self._walked = getattr(self, "_walked", 0) + 1
            if self._walked in (1, 2, 4) and os.environ.get("UZ_WALK_RESERVE", "1") != "0":
                self._ck(self.L.uz_walk_reserve(self.h, 4), "uz_walk_reserve")
            if names:
                kb.qnames = io_native.DeviceNames(self, rid.value, int(kb.n_qnames))
            return rid.value
        nb = int(kb.n_name_bytes) if names else 0
        alloc = getattr(kb, "_alloc", None)
        buf = (alloc(max(1, nb)) if alloc is not None else np.empty(max(1, nb), np.uint8)) if names else None
        self._ck(self.L.uz_reads_from_bam(self.h, int(kb.token), kb.kept.ctypes.data, int(kb.n), kb.aux.ctypes.data, int(kb.n_aux), kb.contig_off.ctypes.data,
                                          kb.max_span.ctypes.data, int(kb.n_contigs), int(kb.n_cigar_total), int(kb.n_row_units), int(kb.n_seq_units),
                                          int(kb.n_qnames), int(kb.min_base_qual), buf.ctypes.data if names else None, nb, C.byref(rid)), "uz_reads_from_bam")
        kb.token = None
        if names:
            kb.set_names(buf[:nb])
        return rid.value

    def adopt_sites(self, view: abi.SitesView) -> int:
        sid = C.c_int(-1)
        self._ck(self.L.uz_sites_adopt_device(self.h, C.byref(view), C.byref(sid)), "uz_sites_adopt_device")
        return sid.value

    def adopt_family(self, sites_h: int, view: abi.FamilyView) -> int:
        fid = C.c_int(-1)
        self._ck(self.L.uz_family_adopt_device(self.h, int(sites_h), C.byref(view), C.byref(fid)), "uz_family_adopt_device")
        return fid.value

    def adopt_reads(self, view: abi.ReadsPackedView) -> int:
        rid = C.c_int(-1)
        self._ck(self.L.uz_reads_adopt_device(self.h, C.byref(view), C.byref(rid)), "uz_reads_adopt_device")
        return rid.value

    def free_sites(self, sid: int):
        self._ck(self.L.uz_sites_free(self.h, int(sid)), "uz_sites_free")
        self._staged_sites.pop(int(sid), None)

    def free_reads(self, rid: int):
        self._ck(self.L.uz_reads_free(self.h, int(rid)), "uz_reads_free")
        self._staged.pop(int(rid), None)

    def drop_derived(self):
        self._ck(self.L.uz_drop_derived(self.h), "uz_drop_derived")

    def sync(self):
        self._ck(self.L.uz_sync(self.h), "uz_sync")

    # ----------------------------------------------------------- site stage
    def site_scan(self, fam: int):
        self._ck(self.L.uz_site_scan(self.h, int(fam)), "uz_site_scan")

    def site_scan_many(self, fams):
        """cohort form: all the families (of one sites table) in one launch"""
        ids = np.ascontiguousarray(np.asarray(list(fams), dtype=np.int32))
        self._ck(self.L.uz_site_scan_many(self.h, ids.ctypes.data, int(ids.size)), "uz_site_scan_many")

    def classify(self, fam: int, params: abi.Params, n_sites: int) -> np.ndarray:
        self.set_params(params)
        out = np.zeros(max(1, n_sites), dtype=np.uint8)
        self._ck(self.L.uz_site_classes(self.h, int(fam), out.ctypes.data), "uz_site_classes")
        return out[:n_sites]

    def find(self, fam: int, dv: abi.Held, params: abi.Params, mode: int, fetch: bool = True, transient: bool = False):
        """transient: the index lists come back as views of page-locked buffers this engine re-uses (valid until the find after next) -- what a
        pipeline that reads them at once wants (pipeline.run_pipelined)"""
        self.set_params(params)
        n = dv.view.n
        co = np.zeros(n + 1, dtype=np.int64)
        ho = np.zeros(n + 1, dtype=np.int64)
        self._ck(self.L.uz_find(self.h, int(fam), dv.ref(), int(mode), co.ctypes.data, ho.ctypes.data), "uz_find")
        if not fetch:
            return co, None, None, ho, None
        ci, cf, hi = self._fetch(int(co[n]), int(ho[n]), transient=transient)
        return co, ci, cf, ho, hi

    def _fetch(self, nc: int, nh: int, transient: bool = False):
        if transient:
            # Into page-locked memory kept from call to call (three sets in turn: the lists of a find are read before the find after next
            # returns): the library then copies by kernels, past the DMA engine's queue -- inside a staged pass a plain copy back waits
            # behind the records of the chunk before (abi.hip: uz_find_fetch)
            if not hasattr(self, "_find_pins"):
                self._find_pins, self._find_turn = [None, None, None], 0
            self._find_turn = (self._find_turn + 1) % 3
            need = 4 * max(1, nc) + 256 + max(1, nc) + 256 + 4 * max(1, nh) + 256
            pin = self._find_pins[self._find_turn]
            if pin is None or pin[1].size < need:
                if pin is not None:
                    pin[0].free_all()
                pool = PinnedPool()
                pin = self._find_pins[self._find_turn] = (pool, pool.alloc(need + need // 4 + (1 << 16)))
            buf = pin[1]
            a = (4 * max(1, nc) + 255) & ~255
            b = a + ((max(1, nc) + 255) & ~255)
            ci, cf, hi = buf[: 4 * max(1, nc)].view(np.int32), buf[a: a + max(1, nc)], buf[b: b + 4 * max(1, nh)].view(np.int32)
        else:
            ci = np.zeros(max(1, nc), dtype=np.int32)
            cf = np.zeros(max(1, nc), dtype=np.uint8)
            hi = np.zeros(max(1, nh), dtype=np.int32)
        self._ck(self.L.uz_find_fetch(self.h, ci.ctypes.data, cf.ctypes.data, hi.ctypes.data), "uz_find_fetch")
        return ci[:nc], cf[:nc], hi[:nh]

    # ----------------------------------------------------------- read stage
    def phase_raw(self, fam: int, reads_h: int, dv: abi.Held, params: abi.Params, find_mode: int):
        self.set_params(params)
        n = dv.view.n
        status = np.empty(max(1, n), dtype=np.int32)  # (the call writes all n entries of each, or raises)
        counts = np.empty(max(1, 4 * n), dtype=np.int32)
        origin = np.empty(max(1, n), dtype=np.int32)
        evidence = np.empty(max(1, n), dtype=np.int32)
        self._ck(
            self.L.uz_phase(self.h, int(fam), int(reads_h), dv.ref(), int(find_mode), status.ctypes.data,
                            counts.ctypes.data, origin.ctypes.data, evidence.ctypes.data),
            "uz_phase",
        )
        return dict(status=status[:n], counts=counts[: 4 * n].reshape(n, 4), origin=origin[:n], evidence=evidence[:n])

    def phase_begin(self, fam: int, reads_h: int, dv: abi.Held, params: abi.Params, find_mode: int):
        """first half of phase_raw (uz_phase_begin): queues the batch and returns; phase_end -- same arguments -- hands out the results.
        Uploads and find() of OTHER batches may run in between: they queue up behind this batch's kernels."""
        self.set_params(params)
        self._ck(self.L.uz_phase_begin(self.h, int(fam), int(reads_h), dv.ref(), int(find_mode)), "uz_phase_begin")

    def phase_end(self, fam: int, reads_h: int, dv: abi.Held, params: abi.Params, find_mode: int):
        n = dv.view.n
        status = np.empty(max(1, n), dtype=np.int32)
        counts = np.empty(max(1, 4 * n), dtype=np.int32)
        origin = np.empty(max(1, n), dtype=np.int32)
        evidence = np.empty(max(1, n), dtype=np.int32)
        self._ck(self.L.uz_phase_end(self.h, int(fam), int(reads_h), dv.ref(), int(find_mode), status.ctypes.data, counts.ctypes.data,
                                     origin.ctypes.data, evidence.ctypes.data), "uz_phase_end")
        return dict(status=status[:n], counts=counts[: 4 * n].reshape(n, 4), origin=origin[:n], evidence=evidence[:n])

    def phase_cohort(self, groups, dv: abi.Held, params: abi.Params, found_list=None, want_lists: bool = True,
                     find_mode: int = abi.FIND_SECOND_WINDOW):
        """Cohort form of phase(): groups = [(fam, reads_h, first, count, cutoff)] over the DNMs of `dv` (several kids, each
        with its own family columns / alignment records / insert cutoff), one launch sequence.  Same result layout as
        phase(); the query-name ids of the lists are those of each group's own table."""
        self.set_params(params)
        n = dv.view.n
        arr = (abi.CohortGroup * len(groups))()
        for k, (fam, rh, first, count, cutoff) in enumerate(groups):
            arr[k].fam_id, arr[k].reads_id, arr[k].dnm_first, arr[k].dnm_count, arr[k].cutoff = int(fam), int(rh), int(first), int(count), float(cutoff)
        status = np.zeros(max(1, n), dtype=np.int32)
        counts = np.zeros(max(1, 4 * n), dtype=np.int32)
        origin = np.zeros(max(1, n), dtype=np.int32)
        evidence = np.zeros(max(1, n), dtype=np.int32)
        self._ck(self.L.uz_phase_cohort(self.h, arr, len(groups), dv.ref(), int(find_mode), status.ctypes.data, counts.ctypes.data,
                                        origin.ctypes.data, evidence.ctypes.data), "uz_phase_cohort")
        r = dict(status=status[:n], counts=counts[: 4 * n].reshape(n, 4), origin=origin[:n], evidence=evidence[:n], lists=None)
        if want_lists:
            vo, vv = self.votes(n)
            r["lists"] = [tuple(vv[vo[4 * k + j]: vo[4 * k + j + 1]] for j in range(4)) for k in range(n)]
        return r

    def votes(self, n: int):
        vo = np.zeros(4 * n + 1, dtype=np.int64)
        self._ck(self.L.uz_phase_votes(self.h, vo.ctypes.data, None), "uz_phase_votes")
        vv = np.zeros(max(1, int(vo[-1])), dtype=np.int32)
        self._ck(self.L.uz_phase_votes(self.h, vo.ctypes.data, vv.ctypes.data), "uz_phase_votes")
        return vo, vv[: int(vo[-1])]

    def groups(self, n: int):
        go = np.zeros(2 * n + 1, dtype=np.int64)
        self._ck(self.L.uz_phase_groups(self.h, go.ctypes.data, None), "uz_phase_groups")
        gq = np.zeros(max(1, int(go[-1])), dtype=np.int32)
        self._ck(self.L.uz_phase_groups(self.h, go.ctypes.data, gq.ctypes.data), "uz_phase_groups")
        return go, gq[: int(go[-1])]

    def phase(self, fam: int, reads_h: int, dv: abi.Held, params: abi.Params, found_list, want_lists: bool = True,
              find_mode: int = abi.FIND_SECOND_WINDOW):
        """Backend entry used by PhasingHost.run_read_phasing (found_list is ignored:
        the window lists are recomputed on the device for the batch)."""
        r = self.phase_raw(fam, reads_h, dv, params, find_mode)
        n = dv.view.n
        if want_lists:
            vo, vv = self.votes(n)
            r["lists"] = [tuple(vv[vo[4 * k + j]: vo[4 * k + j + 1]] for j in range(4)) for k in range(n)]
        else:
            r["lists"] = None
        return r

    # ------------------------------------------------------------ CNV stage
    def phase_cnv(self, fam: int, dv: abi.Held, params: abi.Params, rb_counts=None, want_lists: bool = True):
        """K6: allele-balance phasing of the DEL / DUP of the batch + summarize_record's decision (merged with the
        read-backed counts when given).  -> dict(cnv_counts [n,2], origin, evidence, etype[, lists: (dad, mom) positions])"""
        self.set_params(params)
        n = dv.view.n
        cnt = np.zeros(max(1, 2 * n), np.int32)
        origin = np.zeros(max(1, n), np.int32)
        evidence = np.zeros(max(1, n), np.int32)
        etype = np.zeros(max(1, n), np.int32)
        rb = None
        if rb_counts is not None:
            rb = np.ascontiguousarray(rb_counts, np.int32).reshape(-1)
            assert rb.size == 4 * n
        self._ck(self.L.uz_phase_cnv(self.h, int(fam), dv.ref(), rb.ctypes.data if rb is not None else None, cnt.ctypes.data,
                                     origin.ctypes.data, evidence.ctypes.data, etype.ctypes.data), "uz_phase_cnv")
        r = dict(cnv_counts=cnt[: 2 * n].reshape(n, 2), origin=origin[:n], evidence=evidence[:n], etype=etype[:n], lists=None)
        if want_lists:
            off = np.zeros(2 * n + 1, np.int64)
            self._ck(self.L.uz_phase_cnv_sites(self.h, off.ctypes.data, None), "uz_phase_cnv_sites")
            pos = np.zeros(max(1, int(off[-1])), np.int32)
            self._ck(self.L.uz_phase_cnv_sites(self.h, off.ctypes.data, pos.ctypes.data), "uz_phase_cnv_sites")
            r["lists"] = [(pos[off[2 * k]: off[2 * k + 1]], pos[off[2 * k + 1]: off[2 * k + 2]]) for k in range(n)]
        return r

    # ---------------------------------------------------------- measurement
    def prof_enable(self, on=True):
        """on: False / True, or the kernel ids (abi.K_*) to time -- an empty list is False"""
        v = (1 if on else 0) if isinstance(on, (bool, int)) else sum(1 << (int(k) + 1) for k in set(on))
        self._ck(self.L.uz_prof_enable(self.h, int(v)), "uz_prof_enable")

    def prof_reset(self):
        self._ck(self.L.uz_prof_reset(self.h), "uz_prof_reset")

    def prof_get(self, kernel: int):
        ms = C.c_double(0)
        n = C.c_int64(0)
        self._ck(self.L.uz_prof_get(self.h, int(kernel), C.byref(ms), C.byref(n)), "uz_prof_get")
        return ms.value, n.value

    def prof_units(self, kernel: int) -> int:
        u = C.c_int64(0)
        self._ck(self.L.uz_prof_units(self.h, int(kernel), C.byref(u)), "uz_prof_units")
        return int(u.value)


class PinnedPair:
    """The two page-locked buffers a staged batch needs when the device inflates its blocks (gathered bytes, inflated bytes), kept and
    grown from batch to batch: `alloc` is BamSource.select's inflate_alloc (called twice per batch, after start())."""

    def __init__(self):
        self.pool = PinnedPool()
        self.bufs = []
        self.k = 0

    def start(self):
        self.k = 0

    def alloc(self, nbytes: int) -> np.ndarray:
        i, self.k = self.k, self.k + 1
        while len(self.bufs) <= i:
            self.bufs.append(None)
        if self.bufs[i] is None or self.bufs[i].size < nbytes:
            self.bufs[i] = self.pool.alloc(int(nbytes) + int(nbytes) // 4 + (1 << 20))  # (the block it outgrew stays until free_all)
        return self.bufs[i][: max(64, int(nbytes))]

    def free_all(self):
        self.bufs = []
        self.pool.free_all()


class PinnedPool:
    """Page-locked host memory for the staged columns (uz_pinned_alloc), handed out as numpy uint8 arrays.
    alloc() is the `alloc` callback of abi.packed_view_alloc / io_native.pack_reads / ReadsSource.select."""

    def __init__(self):
        self.L = load_library()
        self._blocks = []
        self._slab = None  # [base address, capacity, used]: alloc() carves from it while it lasts
        self._ended = None  # the slab end_slab() closed (rewind() takes it up again)
        self.keep = False   # kept from batch to batch by its owner (io_native.BamSource.select rewinds it instead of pinning a new block)
        self.last_slab_used = 0

    def new_slab(self, nbytes: int):
        """The allocations that follow come back to back (256-byte aligned) out of ONE page-locked block of `nbytes`: the
        columns of a table staged this way cross the link as one copy (abi.hip: SlabPlan).  What does not fit any more gets
        a block of its own, as without a slab."""
        nbytes = max(1 << 20, int(nbytes))
        p = C.c_void_p()
        if self.L.uz_pinned_alloc(nbytes, C.byref(p)) != 0 or not p.value:
            raise MemoryError("uz_pinned_alloc(%d) failed" % nbytes)
        self._blocks.append(p.value)
        self.last_slab_used = self._slab[2] if self._slab else 0
        self._slab = [p.value, nbytes, 0]

    def rewind(self, nbytes: int) -> bool:
        """Start over in the current slab when it holds at least `nbytes` (a staging loop re-uses its page-locked block chunk after
        chunk instead of pinning a new one); the overflow blocks of the last use are freed.  False: no slab of that size."""
        if self._slab is None and getattr(self, "_ended", None) is not None:
            self._slab = self._ended  # (end_slab() closed it: the block is still there)
        if self._slab is None or self._slab[1] < int(nbytes):
            return False
        keep = self._slab[0]
        for p in self._blocks:
            if p != keep:
                self.L.uz_pinned_free(C.c_void_p(p))
        self._blocks = [keep]
        self._slab[2] = 0
        return True

    def slab_used(self) -> int:
        return self._slab[2] if self._slab else 0

    def end_slab(self) -> int:
        used = self._slab[2] if self._slab else 0
        self._ended = self._slab
        self._slab = None
        return used

    def alloc(self, nbytes: int) -> np.ndarray:
        nbytes = max(64, int(nbytes))
        if self._slab is not None:
            at = (self._slab[2] + 255) & ~255
            if at + nbytes <= self._slab[1]:
                self._slab[2] = at + nbytes
                buf = (C.c_uint8 * nbytes).from_address(self._slab[0] + at)
                return np.frombuffer(buf, dtype=np.uint8, count=nbytes)
        p = C.c_void_p()
        if self.L.uz_pinned_alloc(nbytes, C.byref(p)) != 0 or not p.value:
            raise MemoryError("uz_pinned_alloc(%d) failed" % nbytes)
        self._blocks.append(p.value)
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=np.uint8, count=nbytes)

    def free_all(self):
        for p in self._blocks:
            self.L.uz_pinned_free(C.c_void_p(p))
        self._blocks = []
        self._slab = None
        self._ended = None
