"""Builds and wraps tests/emu/emu_phase.cpp (CPU emulation of the read-stage kernel
body; debugging aid, tests only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_LIB = None


def build():
    so = os.path.join(_HERE, "libemu_phase.so")
    deps = [os.path.join(_HERE, "emu_phase.cpp"), os.path.join(_ROOT, "unfazed_amd", "csrc", "phase_body.hpp"),
            os.path.join(_ROOT, "unfazed_amd", "csrc", "wg.hpp"), os.path.join(_ROOT, "unfazed_amd", "csrc", "pack.hpp"),
            os.path.join(_ROOT, "include", "uz_types.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
                               "-Wno-unused-function", "-Wno-unused-variable",
                               "-I", os.path.join(_ROOT, "include"), "-I", os.path.join(_ROOT, "unfazed_amd", "csrc"),
                               deps[0], "-o", so])
    return so


def phase(params, sites, reads, dnms, found, no_seq=None, umask=None, bl=None):
    """Same result layout as oracle.phase(..., keep_lists=True) plus groups."""
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.emu_phase.restype = C.c_int
    co, ci, cf, ho, hi = found
    n = dnms.view.n
    ci = np.ascontiguousarray(ci if ci.size else np.zeros(1, np.int32))
    cf = np.ascontiguousarray(cf if cf.size else np.zeros(1, np.uint8))
    hi = np.ascontiguousarray(hi if hi.size else np.zeros(1, np.int32))
    status = np.zeros(max(1, n), np.int32)
    counts = np.zeros(max(1, 4 * n), np.int32)
    origin = np.zeros(max(1, n), np.int32)
    evidence = np.zeros(max(1, n), np.int32)
    lstart = np.zeros(max(1, n), np.int64)
    llen = np.zeros(max(1, 6 * n), np.int32)
    cap = 1 << 22
    pool = np.zeros(cap, np.int32)
    used = C.c_longlong(0)
    base_err = C.c_int32(0)
    if no_seq is not None:
        no_seq = np.ascontiguousarray(no_seq, np.uint8)
    if umask is not None:
        umask = np.ascontiguousarray(umask, np.uint16)
    bl_off = bl_pos = None
    if bl is not None:  # (offsets [n + 1], positions): the listed bases of every record (with umask)
        bl_off, bl_pos = np.ascontiguousarray(bl[0], np.int64), np.ascontiguousarray(bl[1], np.uint16)
    vp = C.c_void_p
    rc = _LIB.emu_phase(C.byref(params), sites.ref(), reads.ref(), dnms.ref(), vp(co.ctypes.data), vp(ci.ctypes.data),
                        vp(cf.ctypes.data), vp(ho.ctypes.data), vp(hi.ctypes.data), vp(status.ctypes.data),
                        vp(counts.ctypes.data), vp(origin.ctypes.data), vp(evidence.ctypes.data),
                        vp(lstart.ctypes.data), vp(llen.ctypes.data), vp(pool.ctypes.data), C.c_longlong(cap),
                        C.byref(used), vp(no_seq.ctypes.data) if no_seq is not None else None, C.byref(base_err),
                        vp(umask.ctypes.data) if umask is not None else None,
                        vp(bl_off.ctypes.data) if bl_off is not None else None, vp(bl_pos.ctypes.data) if bl_pos is not None else None)
    assert rc == 0 and used.value <= cap
    llen = llen[: 6 * n].reshape(n, 6)
    lists = []
    for d in range(n):
        o = int(lstart[d])
        row = []
        for k in range(6):
            ln = int(llen[d, k]) if o >= 0 else 0
            row.append(pool[o: o + ln].copy() if o >= 0 else np.zeros(0, np.int32))
            if o >= 0:
                o += ln
        lists.append(row)
    return dict(status=status[:n], counts=counts[: 4 * n].reshape(n, 4), origin=origin[:n], evidence=evidence[:n],
                lists=lists, base_err=int(base_err.value))
