"""Python side of the benchmark-scale generator (synth/uzsynth.h): builds the two generator libraries and exposes
  * reads_cpu(...)     -> the ASCII columns (uz_reads_view) of a cluster range, generated on the host (for the oracle)
  * WorkloadOnGpu(...) -> sites, family and the whole reads table generated in place in HBM in the staged format
                          (uz_reads_packed_view; device pointers)
Test / bench infrastructure only."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

from unfazed_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
READLEN, ROW, MAXOPS, UNITS, MAXSEG = 151, 160, 3, 5, 16384
HALF_WIDTH = 6000


def _build(target, cmd):
    deps = [os.path.join(_HERE, "uzsynth.h")] + [c for c in cmd if c.endswith((".c", ".hip", ".cpp"))]

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


def build_files():
    """the file writer (synth/uzfiles.cpp): BAM + BAI, BGZF VCF + TBI"""
    return _build(os.path.join(_HERE, "libuzsynth_files.so"),
                  ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", _HERE, os.path.join(_HERE, "uzfiles.cpp"), "-lz", "-ldl"])


class Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_clusters", C.c_int32), ("min_base_qual", C.c_int32)]


class SitesS(C.Structure):
    _fields_ = [("contig_off", C.c_void_p), ("pos", C.c_void_p), ("ref_base", C.c_void_p), ("alt_base", C.c_void_p),
                ("khap", C.c_void_p), ("n_contigs", C.c_int32)]


class DnmsS(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("site_idx", C.c_void_p), ("kind", C.c_void_p), ("len", C.c_void_p), ("origin", C.c_void_p)]


class ClustersS(C.Structure):
    _fields_ = [("contig", C.c_void_p), ("lo", C.c_void_p), ("hi", C.c_void_p), ("d0", C.c_void_p), ("nd", C.c_void_p),
                ("pair_off", C.c_void_p), ("cigar_off", C.c_void_p)]


# packed output (GPU): per-record columns, then cigar / seq2 (two-bit base rows) / qlow
PACKED_COLS = abi.PACKED_RECORD_COLS


class OutPackedS(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("start", "end", "tlen", "mate", "qname", "flag", "l_seq", "n_cigar", "mapq", "aux",
                                          "cigar", "seq2", "qlow")]


ASCII_COLS = [("start", np.int32, 1), ("end", np.int32, 1), ("flag", np.uint16, 1), ("mapq", np.uint8, 1),
              ("aux", np.uint8, 1), ("tlen", np.int32, 1), ("qname", np.uint32, 1), ("mate", np.int32, 1),
              ("cigar_off", np.uint32, 1), ("n_cigar", np.uint16, 1), ("cigar", np.uint32, 0),
              ("l_seq", np.uint16, 1), ("sq_off16", np.uint32, 1), ("seq", np.uint8, ROW), ("qual", np.uint8, ROW)]


class OutAsciiS(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _, _ in ASCII_COLS]


def make_cfg(seed=203, n_clusters=0, min_base_qual=20):
    c = Cfg()
    c.seed, c.n_clusters, c.min_base_qual = seed, n_clusters, min_base_qual
    return c


def reads_contig_off(cl, c0, c1, n_contigs):
    """contig_off of the reads table made of clusters [c0, c1) (records relative to cluster c0)"""
    recs = 2 * np.diff(cl.pair_off[c0: c1 + 1])
    cnt = np.bincount(cl.contig[c0:c1], weights=recs, minlength=n_contigs).astype(np.int64)
    off = np.zeros(n_contigs + 1, dtype=np.int64)
    off[1:] = np.cumsum(cnt)
    return off


def _host_structs(cfg, sc, dn, cl):
    keep = []

    def ptr(a, dt):
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return a.ctypes.data
    S = SitesS()
    S.contig_off, S.pos = ptr(sc.contig_off, np.int64), ptr(sc.pos, np.int32)
    S.ref_base, S.alt_base, S.khap = ptr(sc.ref_base, np.uint8), ptr(sc.alt_base, np.uint8), ptr(sc.khap, np.uint8)
    S.n_contigs = len(sc.contig_off) - 1
    D = DnmsS()
    D.pos, D.site_idx = ptr(dn.start, np.int32), ptr(dn.site_idx, np.int32)
    D.kind, D.len, D.origin = ptr(dn.kind, np.uint8), ptr(dn.length, np.uint8), ptr(dn.origin, np.uint8)
    K = ClustersS()
    K.contig, K.lo, K.hi = ptr(cl.contig, np.int32), ptr(cl.lo, np.int32), ptr(cl.hi, np.int32)
    K.d0, K.nd, K.pair_off = ptr(cl.d0, np.int32), ptr(cl.nd, np.int32), ptr(cl.pair_off, np.int64)
    K.cigar_off = 0
    return S, D, K, keep


def reads_cpu(cfg, sc, dn, cl, c0, c1, threads=1):
    """Generate the record blocks of clusters [c0, c1) on the host in the ASCII form.  Returns (abi.Held view, dict of
    arrays).  Record indices / qname ids / CIGAR offsets are relative to cluster c0."""
    L = C.CDLL(build_cpu())
    L.uzs_gen_reads_cpu.restype = C.c_int
    L.uzs_count_ops_cpu.restype = C.c_int64
    S, D, K, keep = _host_structs(cfg, sc, dn, cl)
    n = int(2 * (cl.pair_off[c1] - cl.pair_off[c0]))
    # slices of clusters, one worker each (ctypes releases the GIL)
    threads = max(1, min(threads, c1 - c0))
    cuts = [c0 + (c1 - c0) * k // threads for k in range(threads + 1)]
    ops = [0] * threads

    def count(k):
        ops[k] = L.uzs_count_ops_cpu(C.byref(cfg), C.byref(K), C.byref(D), C.c_int32(cuts[k]), C.c_int32(cuts[k + 1]))
    th = [threading.Thread(target=count, args=(k,)) for k in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    n_ops = int(sum(ops))
    arrs = {}
    for name, dt, w in ASCII_COLS:
        arrs[name] = np.zeros(max(1, n * w if w else n_ops), dtype=dt)
    rc = [0] * threads

    def gen(k):
        r0 = int(2 * (cl.pair_off[cuts[k]] - cl.pair_off[c0]))
        g0 = int(sum(ops[:k]))
        O = OutAsciiS()
        for name, dt, w in ASCII_COLS:
            off = (r0 * w if w else g0) * np.dtype(dt).itemsize
            setattr(O, name, arrs[name].ctypes.data + off)
        rc[k] = L.uzs_gen_reads_cpu(C.byref(cfg), C.byref(S), C.byref(D), C.byref(K), C.c_int32(cuts[k]), C.c_int32(cuts[k + 1]),
                                    C.byref(O))
    th = [threading.Thread(target=gen, args=(k,)) for k in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(r == 0 for r in rc), rc
    # the workers numbered records / pairs / CIGAR words from their own first cluster: shift to the range's
    for k in range(1, threads):
        r0 = int(2 * (cl.pair_off[cuts[k]] - cl.pair_off[c0]))
        r1 = int(2 * (cl.pair_off[cuts[k + 1]] - cl.pair_off[c0]))
        arrs["mate"][r0:r1] += r0
        arrs["qname"][r0:r1] += r0 // 2
        arrs["cigar_off"][r0:r1] += int(sum(ops[:k]))
        arrs["sq_off16"][r0:r1] += r0 * (ROW // 16)
    nc = len(sc.contig_off) - 1
    arrs["contig_off"] = reads_contig_off(cl, c0, c1, nc)
    arrs["max_span"] = np.full(nc, READLEN + 12, dtype=np.int32)
    v = abi.ReadsView()
    v.n_segs = n
    v.n_contigs = nc
    for name in list(arrs):
        setattr(v, name, arrs[name].ctypes.data)
    v.n_cigar_total = n_ops
    v.n_sq_bytes = n * ROW
    v.n_qnames = n // 2
    return abi.Held(v, arrs), arrs


def qname_of(pair: int) -> str:
    """query name the file writer gives global pair number `pair` (uzs_qname)"""
    return "UZSYN:30X:1:%03d:%07d" % (pair % 997, pair // 997)


def _names_array(names):
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    return arr


def write_bam(path, cfg, sc, dn, cl, c0=0, c1=None, contig_len=None, level=6, tags=True, threads=0, bai=True, filler=0.0, filler_reach=65536):
    """Clusters [c0, c1) as a coordinate-sorted BAM (+ BAI next to it): the records reads_cpu() generates for the same range,
    query names qname_of(global pair).  -> dict(records, raw_bytes, file_bytes, blocks)
    filler: coverage of read pairs laid into the gaps BETWEEN the clusters (and filler_reach bases in front of / behind a contig's first / last
    cluster of the range) -- a file with records everywhere, as a real one has; none of them can be returned by a fetch of the clusters' DNMs."""
    L = C.CDLL(build_files())
    c1 = cl.n if c1 is None else c1
    S, D, K, keep = _host_structs(cfg, sc, dn, cl)
    nc = len(sc.contig_off) - 1
    if contig_len is None:
        from .sites_np import GRCH38_LEN
        contig_len = GRCH38_LEN[:nc] if nc <= len(GRCH38_LEN) else [int(sc.pos.max()) + 100000] * nc
    lens = np.ascontiguousarray(contig_len, np.int32)
    names = _names_array(sc.contig_names)
    stats = (C.c_int64 * 4)()
    err = C.create_string_buffer(256)
    L.uzs_write_bam_filler.restype = C.c_int
    rc = L.uzs_write_bam_filler(os.fsencode(path), os.fsencode(path + ".bai") if bai else None, C.byref(cfg), C.byref(S), C.byref(D), C.byref(K),
                                C.c_int32(c0), C.c_int32(c1), names, C.c_void_p(lens.ctypes.data), C.c_int(level), C.c_int(1 if tags else 0),
                                C.c_int(threads), C.c_double(float(filler)), C.c_int64(int(filler_reach)), stats, err, C.c_int(256))
    if rc != 0:
        raise RuntimeError("uzs_write_bam: " + err.value.decode())
    return dict(zip(("records", "raw_bytes", "file_bytes", "blocks"), (int(x) for x in stats)))


def write_vcf(path, sc, samples=("kid", "dad", "mom"), contig_len=None, level=6, threads=0, tbi=True):
    """The sites table as a BGZF-compressed VCF (+ tabix index next to it).  -> dict(records, raw_bytes, file_bytes, blocks)"""
    L = C.CDLL(build_files())
    nc = len(sc.contig_off) - 1
    if contig_len is None:
        from .sites_np import GRCH38_LEN
        contig_len = GRCH38_LEN[:nc] if nc <= len(GRCH38_LEN) else [int(sc.pos.max()) + 100000] * nc
    lens = np.ascontiguousarray(contig_len, np.int32)
    keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=np.ascontiguousarray(sc.pos, np.int32),
                sflags=np.ascontiguousarray(sc.sflags, np.uint8), ref=np.ascontiguousarray(sc.ref_base, np.uint8),
                alt=np.ascontiguousarray(sc.alt_base, np.uint8), gt=np.ascontiguousarray(sc.gt, np.uint8))
    cols = {k: [np.ascontiguousarray(getattr(sc, k)[m], np.uint16) for m in range(3)] for k in ("rd", "ad", "gq")}
    ptrs = {k: (C.c_void_p * 3)(*[a.ctypes.data for a in v]) for k, v in cols.items()}
    stats = (C.c_int64 * 4)()
    err = C.create_string_buffer(256)
    L.uzs_write_vcf.restype = C.c_int
    rc = L.uzs_write_vcf(os.fsencode(path), os.fsencode(path + ".tbi") if tbi else None, C.c_int64(sc.n), C.c_int32(nc),
                         C.c_void_p(keep["contig_off"].ctypes.data), _names_array(sc.contig_names), C.c_void_p(lens.ctypes.data),
                         C.c_void_p(keep["pos"].ctypes.data), C.c_void_p(keep["sflags"].ctypes.data), C.c_void_p(keep["ref"].ctypes.data),
                         C.c_void_p(keep["alt"].ctypes.data), C.c_void_p(keep["gt"].ctypes.data), ptrs["rd"], ptrs["ad"], ptrs["gq"],
                         _names_array(list(samples)), C.c_int(level), C.c_int(threads), stats, err, C.c_int(256))
    if rc != 0:
        raise RuntimeError("uzs_write_vcf: " + err.value.decode())
    return dict(zip(("records", "raw_bytes", "file_bytes", "blocks"), (int(x) for x in stats)))


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

    def get_into(self, out, ptr):
        if out.nbytes:
            assert self.L.uzs_d2h(out.ctypes.data, ptr, out.nbytes) == 0
        return out

    def free_all(self):
        for p in self.ptrs:
            self.L.uzs_dev_free(p)
        self.ptrs = []


class WorkloadOnGpu:
    """Sites table + family columns + the reads table of all clusters, resident in HBM (staged format).
    c0, c1: only clusters [c0, c1) are generated (one rank's shard of a list every rank knows whole: a cluster's records are a function of
    its number in the WHOLE list, so a shard holds exactly the records the one-GPU table holds for its clusters); record numbers and
    query-name ids are then relative to the first record / pair of cluster c0."""

    def __init__(self, cfg, sc, dn, cl, device=0, c0=0, c1=None):
        self.cfg, self.sc, self.dn, self.cl = cfg, sc, dn, cl
        cfg.n_clusters = cl.n
        c1 = cl.n if c1 is None else int(c1)
        self.c0, self.c1 = int(c0), c1
        dev = self.dev = DeviceArrays(device)
        p0, p1 = int(cl.pair_off[self.c0]), int(cl.pair_off[c1])
        n = int(2 * (p1 - p0))
        self.n_segs = n
        nseg = 2 * np.diff(cl.pair_off)
        if nseg.max() > MAXSEG:
            raise ValueError("a cluster holds %d records (> %d)" % (nseg.max(), MAXSEG))
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
        D.pos = dev.put(np.ascontiguousarray(dn.start, np.int32))
        D.site_idx = dev.put(np.ascontiguousarray(dn.site_idx, np.int32))
        D.kind, D.len, D.origin = dev.put(dn.kind), dev.put(dn.length), dev.put(dn.origin)
        K = ClustersS()
        K.contig, K.lo, K.hi = dev.put(cl.contig), dev.put(cl.lo), dev.put(cl.hi)
        # (offsets relative to the first generated cluster; clusters outside [c0, c1) are never touched)
        K.d0, K.nd, K.pair_off = dev.put(cl.d0), dev.put(cl.nd), dev.put(np.ascontiguousarray(np.clip(cl.pair_off - p0, 0, p1 - p0), np.int64))
        # pass 1: CIGAR words per cluster -> offsets
        d_ops = dev.alloc(8 * max(1, cl.n))
        K.cigar_off = 0
        dev.L.uzs_count_ops_hip.restype = C.c_int
        dev.L.uzs_count_ops_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        if dev.L.uzs_count_ops_hip(C.byref(cfg), C.byref(K), C.byref(D), d_ops) != 0:
            raise RuntimeError("uzs_count_ops_hip failed")
        ops = dev.get(d_ops, (cl.n,), np.int64)
        cigar_off = np.zeros(cl.n + 1, np.int64)
        cigar_off[1:] = np.cumsum(ops)
        cigar_off = np.clip(cigar_off - cigar_off[self.c0], 0, cigar_off[c1] - cigar_off[self.c0])
        self.n_cigar_total = int(cigar_off[c1])
        self.n_row_units = n * UNITS
        K.cigar_off = self._d_cigar_off = dev.put(cigar_off)
        # pass 2: the records
        O = OutPackedS()
        self.out_ptrs = {}
        for name, dt in PACKED_COLS:
            self.out_ptrs[name] = dev.alloc(max(64, n * np.dtype(dt).itemsize + 64))
        self.out_ptrs["cigar"] = dev.alloc(4 * self.n_cigar_total + 64)
        self.out_ptrs["seq2"] = dev.alloc(8 * self.n_row_units + 64)
        self.out_ptrs["qlow"] = dev.alloc(4 * self.n_row_units + 64)
        for name in self.out_ptrs:
            setattr(O, name, self.out_ptrs[name])
        mine = np.zeros(cl.n, bool)
        mine[self.c0: c1] = True
        small = np.nonzero((nseg <= 4096) & mine)[0].astype(np.int32)
        big = np.nonzero((nseg > 4096) & mine)[0].astype(np.int32)
        d_small, d_big = dev.put(small if small.size else np.zeros(1, np.int32)), dev.put(big if big.size else np.zeros(1, np.int32))
        dev.L.uzs_gen_reads_hip.restype = C.c_int
        dev.L.uzs_gen_reads_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                            C.c_int32, C.c_void_p]
        rc = dev.L.uzs_gen_reads_hip(C.byref(cfg), C.byref(S), C.byref(D), C.byref(K), d_small, int(small.size), d_big, int(big.size),
                                     C.byref(O))
        if rc != 0:
            raise RuntimeError("uzs_gen_reads_hip failed: %d" % rc)
        nc = len(sc.contig_off) - 1
        self.contig_off = reads_contig_off(cl, self.c0, c1, nc)
        self.max_span = np.full(nc, READLEN + 12, dtype=np.int32)
        self.d_rcontig_off = dev.put(self.contig_off)
        self.d_max_span = dev.put(self.max_span)

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

    def _fill_view(self, v):
        v.n_segs = self.n_segs
        v.n_contigs = len(self.sc.contig_off) - 1
        v.min_base_qual = self.cfg.min_base_qual
        v.n_cigar_total, v.n_row_units, v.n_seq_units = self.n_cigar_total, self.n_row_units, self.n_row_units
        v.n_qnames = self.n_segs // 2
        return v

    def reads_view(self):
        """uz_reads_packed_view over the DEVICE columns (uz_reads_adopt_device)"""
        v = self._fill_view(abi.ReadsPackedView())
        v.contig_off, v.max_span = self.d_rcontig_off, self.d_max_span
        for name in self.out_ptrs:
            setattr(v, name, self.out_ptrs[name])
        return v

    def download(self, c0=0, c1=None, alloc=None):
        """The records of clusters [c0, c1) copied back to the host as a self-contained packed view (abi.Held): record
        numbers (mate links) are relative to the first record of c0, query-name ids stay global."""
        cl = self.cl
        c1 = self.c1 if c1 is None else c1
        c0 = max(c0, self.c0)
        if not (self.c0 <= c0 <= c1 <= self.c1):
            raise ValueError("clusters [%d, %d) are not in this table ([%d, %d))" % (c0, c1, self.c0, self.c1))
        nc = len(self.sc.contig_off) - 1
        pbase = int(cl.pair_off[self.c0])
        r0, r1 = int(2 * (cl.pair_off[c0] - pbase)), int(2 * (cl.pair_off[c1] - pbase))
        if not hasattr(self, "_cigar_off_h"):
            self._cigar_off_h = self.dev.get(self._d_cigar_off, (cl.n + 1,), np.int64)
        g0, g1 = int(self._cigar_off_h[c0]), int(self._cigar_off_h[c1])
        n = r1 - r0
        h = abi.packed_view_alloc(n, nc, g1 - g0, n * UNITS, alloc, n_exc=0)
        v = h.view
        v.min_base_qual = self.cfg.min_base_qual
        v.n_qnames = self.n_segs // 2
        h.arrays["contig_off"][:] = reads_contig_off(cl, c0, c1, nc)
        h.arrays["max_span"][:] = self.max_span
        for name, dt in PACKED_COLS:
            isz = np.dtype(dt).itemsize
            self.dev.get_into(h.arrays[name][:n], self.out_ptrs[name] + r0 * isz)
        if g1 > g0:
            self.dev.get_into(h.arrays["cigar"][: g1 - g0], self.out_ptrs["cigar"] + 4 * g0)
        if n:
            self.dev.get_into(h.arrays["seq2"][: 8 * n * UNITS], self.out_ptrs["seq2"] + 8 * r0 * UNITS)
            self.dev.get_into(h.arrays["qlow"][: 4 * n * UNITS], self.out_ptrs["qlow"] + 4 * r0 * UNITS)
            h.arrays["mate"][:n] -= r0
        return h

    def tlen_head(self, cap=1000001):
        return self.dev.get(self.out_ptrs["tlen"], (min(self.n_segs, cap),), np.int32)

    def free(self):
        self.dev.free_all()
