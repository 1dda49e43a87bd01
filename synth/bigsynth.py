"""Python side of the benchmark-scale generator (synth/uzsynth.h): builds the two
generator libraries and exposes
  * reads_cpu(...)  -> numpy columns of the read blocks of a DNM range (for the oracle)
  * ReadsOnGpu(...) -> the same columns generated in place in HBM (device pointers)
Test / bench infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from unfazed_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
READLEN, ROW, MAXOPS = 151, 160, 3


def _build(target, cmd):
    deps = [os.path.join(_HERE, "uzsynth.h")] + [c for c in cmd if c.endswith((".c", ".hip"))]
    def stale():
        return not os.path.exists(target) or os.path.getmtime(target) < max(os.path.getmtime(d) for d in deps)
    if stale():
        import fcntl
        with open(target + ".lock", "w") as lock:  # ranks of one node may build at once
            fcntl.flock(lock, fcntl.LOCK_EX)
            if stale():
                subprocess.check_call(cmd + ["-o", target + ".tmp"])
                os.replace(target + ".tmp", target)
    return target


def build_cpu():
    return _build(os.path.join(_HERE, "libuzsynth_cpu.so"),
                  ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-I", _HERE, os.path.join(_HERE, "uzsynth_cpu.c")])


def build_hip():
    return _build(os.path.join(_HERE, "libuzsynth_hip.so"),
                  [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                   "-shared", "-I", _HERE, os.path.join(_HERE, "uzsynth_hip.hip")])


class Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_pairs", C.c_int32), ("half_width", C.c_int32), ("n_dnms", C.c_int32),
                ("reserved", C.c_int32)]


class SitesS(C.Structure):
    _fields_ = [("contig_off", C.c_void_p), ("pos", C.c_void_p), ("ref_base", C.c_void_p), ("alt_base", C.c_void_p),
                ("khap", C.c_void_p), ("n_contigs", C.c_int32)]


class DnmsS(C.Structure):
    _fields_ = [("contig", C.c_void_p), ("pos", C.c_void_p), ("site_idx", C.c_void_p), ("kind", C.c_void_p),
                ("len", C.c_void_p), ("origin", C.c_void_p)]


OUT_COLS = [("start", np.int32, 1), ("end", np.int32, 1), ("flag", np.uint16, 1), ("mapq", np.uint8, 1),
            ("aux", np.uint8, 1), ("tlen", np.int32, 1), ("qname", np.uint32, 1), ("mate", np.int32, 1),
            ("cigar_off", np.uint32, 1), ("n_cigar", np.uint16, 1), ("cigar", np.uint32, MAXOPS),
            ("l_seq", np.uint16, 1), ("sq_off16", np.uint32, 1), ("seq", np.uint8, ROW), ("qual", np.uint8, ROW)]


class OutS(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in OUT_COLS]


def make_cfg(seed=203, n_pairs=1200, half_width=6000, n_dnms=0):
    c = Cfg()
    c.seed, c.n_pairs, c.half_width, c.n_dnms = seed, n_pairs, half_width, n_dnms
    return c


def _reads_contig_off(dn_contig, d0, d1, n_contigs, nseg):
    cnt = np.bincount(dn_contig[d0:d1], minlength=n_contigs).astype(np.int64)
    off = np.zeros(n_contigs + 1, dtype=np.int64)
    off[1:] = np.cumsum(cnt) * nseg
    return off


def reads_cpu(cfg, sc, dn, d0, d1):
    """Generate the read blocks of DNMs [d0, d1) on the host.  Returns (abi.Held view, dict of arrays).
    Record indices / qname ids are relative to d0."""
    L = C.CDLL(build_cpu())
    L.uzs_gen_reads_cpu.restype = C.c_int
    nseg = 2 * cfg.n_pairs
    n = (d1 - d0) * nseg
    S = SitesS()
    keep = [np.ascontiguousarray(sc.contig_off, np.int64), np.ascontiguousarray(sc.pos, np.int32),
            np.ascontiguousarray(sc.ref_base), np.ascontiguousarray(sc.alt_base), np.ascontiguousarray(sc.khap)]
    S.contig_off, S.pos, S.ref_base, S.alt_base, S.khap = [a.ctypes.data for a in keep]
    S.n_contigs = len(sc.contig_off) - 1
    D = DnmsS()
    dk = [np.ascontiguousarray(dn.contig, np.int32), np.ascontiguousarray(dn.start, np.int32),
          np.ascontiguousarray(dn.site_idx, np.int32), np.ascontiguousarray(dn.kind, np.uint8),
          np.ascontiguousarray(dn.length, np.uint8), np.ascontiguousarray(dn.origin, np.uint8)]
    D.contig, D.pos, D.site_idx, D.kind, D.len, D.origin = [a.ctypes.data for a in dk]
    O = OutS()
    arrs = {}
    for name, dt, w in OUT_COLS:
        arrs[name] = np.zeros(max(1, n * w), dtype=dt)
        setattr(O, name, arrs[name].ctypes.data)
    rc = L.uzs_gen_reads_cpu(C.byref(cfg), C.byref(S), C.byref(D), C.c_int32(d0), C.c_int32(d1), C.byref(O))
    assert rc == 0
    nc = len(sc.contig_off) - 1
    arrs["contig_off"] = _reads_contig_off(dn.contig, d0, d1, nc, nseg)
    arrs["max_span"] = np.full(nc, READLEN + 12, dtype=np.int32)
    v = abi.ReadsView()
    v.n_segs = n
    v.n_contigs = nc
    for name in list(arrs):
        setattr(v, name, arrs[name].ctypes.data)
    v.n_cigar_total = n * MAXOPS
    v.n_sq_bytes = n * ROW
    v.n_qnames = (d1 - d0) * cfg.n_pairs
    return abi.Held(v, arrs), arrs


class DeviceArrays:
    """Device allocations made through the generator library (hipMalloc)."""

    def __init__(self, device=0):
        self.L = C.CDLL(build_hip())
        self.L.uzs_dev_alloc.restype = C.c_void_p
        self.L.uzs_dev_alloc.argtypes = [C.c_size_t]
        self.L.uzs_dev_free.argtypes = [C.c_void_p]
        self.L.uzs_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.L.uzs_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        assert self.L.uzs_set_device(int(device)) == 0
        self.ptrs = []

    def alloc(self, nbytes):
        p = self.L.uzs_dev_alloc(int(nbytes))
        if not p:
            raise MemoryError("hipMalloc of %d bytes failed" % nbytes)
        self.ptrs.append(p)
        return p

    def put(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.alloc(max(16, a.nbytes))
        if a.nbytes:
            assert self.L.uzs_h2d(p, a.ctypes.data, a.nbytes) == 0
        return p

    def get(self, ptr, shape, dtype):
        out = np.zeros(shape, dtype=dtype)
        if out.nbytes:
            assert self.L.uzs_d2h(out.ctypes.data, ptr, out.nbytes) == 0
        return out

    def free_all(self):
        for p in self.ptrs:
            self.L.uzs_dev_free(p)
        self.ptrs = []


class WorkloadOnGpu:
    """Sites table + family columns + read blocks of all DNMs, resident in HBM."""

    def __init__(self, cfg, sc, dn, device=0):
        self.cfg, self.sc, self.dn = cfg, sc, dn
        dev = self.dev = DeviceArrays(device)
        nseg = 2 * cfg.n_pairs
        n = dn.n * nseg
        self.n_segs = n
        S = SitesS()
        self.d_contig_off = dev.put(np.ascontiguousarray(sc.contig_off, np.int64))
        self.d_pos = dev.put(np.ascontiguousarray(sc.pos, np.int32))
        self.d_sflags = dev.put(sc.sflags)
        self.d_ref = dev.put(sc.ref_base)
        self.d_alt = dev.put(sc.alt_base)
        self.d_khap = dev.put(sc.khap)
        S.contig_off, S.pos, S.ref_base, S.alt_base, S.khap = self.d_contig_off, self.d_pos, self.d_ref, self.d_alt, self.d_khap
        S.n_contigs = len(sc.contig_off) - 1
        self.d_gt = dev.put(sc.gt)
        self.d_rd = [dev.put(sc.rd[m]) for m in range(3)]
        self.d_ad = [dev.put(sc.ad[m]) for m in range(3)]
        self.d_gq = [dev.put(sc.gq[m]) for m in range(3)]
        D = DnmsS()
        D.contig = dev.put(np.ascontiguousarray(dn.contig, np.int32))
        D.pos = dev.put(np.ascontiguousarray(dn.start, np.int32))
        D.site_idx = dev.put(np.ascontiguousarray(dn.site_idx, np.int32))
        D.kind = dev.put(dn.kind)
        D.len = dev.put(dn.length)
        D.origin = dev.put(dn.origin)
        O = OutS()
        self.out_ptrs = {}
        for name, dt, w in OUT_COLS:
            self.out_ptrs[name] = dev.alloc(max(16, n * w * np.dtype(dt).itemsize))
            setattr(O, name, self.out_ptrs[name])
        dev.L.uzs_gen_reads_hip.restype = C.c_int
        rc = dev.L.uzs_gen_reads_hip(C.byref(cfg), C.byref(S), C.byref(D), C.c_int32(0), C.c_int32(dn.n), C.byref(O))
        if rc != 0:
            raise RuntimeError("uzs_gen_reads_hip failed: %d" % rc)
        nc = len(sc.contig_off) - 1
        self.d_rcontig_off = dev.put(_reads_contig_off(dn.contig, 0, dn.n, nc, nseg))
        self.d_max_span = dev.put(np.full(nc, READLEN + 12, dtype=np.int32))

    def sites_view(self):
        v = abi.SitesView()
        v.n_sites = self.sc.n
        v.n_contigs = len(self.sc.contig_off) - 1
        v.contig_off, v.pos, v.sflags, v.ref_base, v.alt_base = self.d_contig_off, self.d_pos, self.d_sflags, self.d_ref, self.d_alt
        return v

    def family_view(self):
        v = abi.FamilyView()
        v.gt = self.d_gt
        for m in range(3):
            v.ref_depth[m], v.alt_depth[m], v.gq[m] = self.d_rd[m], self.d_ad[m], self.d_gq[m]
        return v

    def reads_view(self):
        v = abi.ReadsView()
        v.n_segs = self.n_segs
        v.n_contigs = len(self.sc.contig_off) - 1
        v.contig_off, v.max_span = self.d_rcontig_off, self.d_max_span
        for name, _, _ in OUT_COLS:
            setattr(v, name, self.out_ptrs[name])
        v.n_cigar_total = self.n_segs * MAXOPS
        v.n_sq_bytes = self.n_segs * ROW
        v.n_qnames = self.dn.n * self.cfg.n_pairs
        return v

    def download_block(self, d0, d1):
        """Columns of DNM blocks [d0, d1) copied back to the host (tests: GPU vs CPU generator)."""
        nseg = 2 * self.cfg.n_pairs
        out = {}
        for name, dt, w in OUT_COLS:
            isz = np.dtype(dt).itemsize
            out[name] = self.dev.get(self.out_ptrs[name] + d0 * nseg * w * isz, ((d1 - d0) * nseg * w,), dt)
        return out

    def free(self):
        self.dev.free_all()
