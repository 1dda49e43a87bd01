"""Per-DNM working-set sizes of k_phase on a sample of the bench workload, from the CPU twin of the kernel body
(tests/emu, built with -DUZ_EMU_STATS): candidates, het sites, fetched records, registrations, seeds, pair-table
entries, pairs.  Used to size the LDS arena.  usage: python scripts/phase_sizes.py [n_dnms]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as orc  # noqa: E402
from synth import bigsynth  # noqa: E402
from synth.sites_np import make_clusters, make_sites, place_dnms_full  # noqa: E402
from unfazed_amd import abi  # noqa: E402
from unfazed_amd.hostpath import concordant_cutoff  # noqa: E402

n_sample = int(sys.argv[1]) if len(sys.argv) > 1 else 400
n_sites = int(os.environ.get("UZ_SITES", 20000000))
n_dnms = int(os.environ.get("UZ_DNMS", 100000))
sc = make_sites(n_sites, seed=202)
ev = place_dnms_full(sc, n_dnms, seed=201)
cl = make_clusters(ev)
cfg = bigsynth.make_cfg(seed=203)
c_hi = cl.of_dnm(n_sample - 1) + 1
m = int(cl.d0[c_hi - 1] + cl.nd[c_hi - 1])
rh, _ = bigsynth.reads_cpu(cfg, sc, ev, cl, 0, c_hi, threads=8)
P = abi.make_params()
sv = abi.SitesView()
keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=sc.pos, sflags=sc.sflags, ref_base=sc.ref_base, alt_base=sc.alt_base)
sv.n_sites, sv.n_contigs = sc.n, len(sc.contig_off) - 1
for k, a in keep.items():
    setattr(sv, k, a.ctypes.data)
sh = abi.Held(sv, keep)
fh = abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)
cutoff = concordant_cutoff(np.asarray(rh.arrays["tlen"][:1000000]), P.readlen, 3)
dv = abi.dnms_view(ev.contig[:m], ev.contig[:m], ev.start[:m], ev.end[:m], np.zeros(m, np.uint8), ev.refs[:m], ev.alts[:m], cutoff)
found = orc.find(P, sh, fh, dv, abi.FIND_SECOND_WINDOW)

emu_dir = os.path.join(ROOT, "tests", "emu")
so = os.path.join(emu_dir, "libemu_phase_stats.so")
subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DUZ_EMU_STATS", "-I", os.path.join(ROOT, "include"),
                       "-I", os.path.join(ROOT, "unfazed_amd", "csrc"), os.path.join(emu_dir, "emu_phase.cpp"), "-o", so])
from emu import emu  # noqa: E402
emu._LIB = C.CDLL(so)
emu._LIB.emu_phase.restype = C.c_int
log = np.zeros((m, 12), np.int64)
C.c_void_p.in_dll(emu._LIB, "uz_emu_log").value = log.ctypes.data
r = emu.phase(P, sh, rh, dv, found)
names = ["nc", "nh", "nA", "T", "nI", "E", "S", "M", "P", "Wmax", "levels", "arena_peak_bytes"]
ok = log[:, 7] > 0
print("DNMs %d, reaching the pair table %d" % (m, ok.sum()))
for i, nme in enumerate(names):
    v = log[ok, i]
    print("%-3s mean %8.1f  p50 %6d  p90 %6d  p99 %6d  max %6d" % (nme, v.mean(), *np.percentile(v, [50, 90, 99]).astype(int), v.max()))
np.save("/tmp/phase_sizes.npy", log)
# the host's arena estimate (k_reads.hip: phase_exact_sizes) is a line in the records the het-site fetches return (T, known from the sizing pass)
T, pk = log[ok, 3].astype(float), log[ok, 11].astype(float)
A = np.vstack([T, np.ones_like(T)]).T
coef, *_ = np.linalg.lstsq(A, pk, rcond=None)
res = pk - A @ coef
print("arena peak ~ %.2f * T + %.0f bytes; residual p50 %.0f p99 %.0f max %.0f" % (coef[0], coef[1], *np.percentile(res, [50, 99]), res.max()))
