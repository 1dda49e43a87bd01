import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from test_upload_forms_gpu import _odd_table
from unfazed_amd import abi, io_native
from unfazed_amd.engine import HipEngine, UnfazedHipError
e = HipEngine(0)
rh, arrs, N = _odd_table()
pk = io_native.pack_reads(rh, 20, with_end=True)
src = io_native.ReadsSource(pk)
contig_of = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
fc = np.unique(contig_of).astype(np.int32)
everything = (fc, np.zeros(fc.size, np.int32), np.full(fc.size, 2 ** 31 - 1, np.int32))
lo = arrs["start"][:N][::53].astype(np.int32)
some = (contig_of[::53].astype(np.int32), lo, lo + 1)
for nm, fetches in (("all", everything), ("some", some)):
    for kw in (dict(d16=False), dict(start8=False), dict(pair8=False, narrow8=False), dict(pair8=False), dict()):
        part, idx = src.select(*fetches, want_index=True, **kw)
        for mode in ("host", "own"):
            if mode == "own":
                part.view.pk_sums = None
            try:
                rid = e.upload_reads_packed(part)
                e.reads_headers(rid, idx.size)
                e.free_reads(rid)
                print(nm, kw, mode, "ok", idx.size, sorted(k for k in part.arrays if k.startswith(("tup", "umask", "n_low", "bl")))[:3])
            except UnfazedHipError as x:
                print(nm, kw, mode, "FAIL", str(x)[:80], idx.size)
