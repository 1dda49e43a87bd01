"""BGZF inflate on the device against the host (GPU box): the generator's BAM of N DNMs (30x pile-ups, deflate level 6), every block
through uz_bgzf_inflate (kernel time by HIP events, data resident) and through the host library's inflate (all threads).
    python scripts/inflate_probe.py [N=3000]"""
import gzip
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from synth import bigsynth  # noqa: E402
from synth.sites_np import make_clusters, make_sites, place_dnms_full  # noqa: E402
from unfazed_amd import io_native  # noqa: E402
from unfazed_amd.engine import HipEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
d = tempfile.mkdtemp(prefix="uzinf_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    sc = make_sites(20_000_000, seed=202)
    dn = place_dnms_full(sc, 100000, seed=201)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=203)
    cfg.n_clusters = cl.n
    c_hi = cl.of_dnm(N - 1) + 1
    bam = os.path.join(d, "kid.bam")
    st = bigsynth.write_bam(bam, cfg, sc, dn, cl, 0, c_hi, level=6)
    data = np.fromfile(bam, np.uint8)
    print("BAM: %d records, %.1f MB -> %.1f MB, %d blocks" % (st["records"], data.size / 1e6, st["raw_bytes"] / 1e6, st["blocks"]), flush=True)
    e = HipEngine(0)
    got, nb, ms = e.bgzf_inflate(data, repeat=5)
    print("device: %d blocks, %.2f ms per launch = %.1f GB/s of output (%.1f GB/s of input)" % (nb, ms, got.size / ms / 1e6, data.size / ms / 1e6), flush=True)
    # the working form: pinned buffers, both copies inside the call (uz_bgzf_inflate_to_host), as a staged batch uses it
    from unfazed_amd.engine import PinnedPair
    pair = PinnedPair()
    blocks = []
    at = 0
    while at + 18 <= data.size:
        bsize = (int(data[at + 16]) | (int(data[at + 17]) << 8)) + 1
        blocks.append((at + 18, int(data[at + bsize - 4]) | (int(data[at + bsize - 3]) << 8) | (int(data[at + bsize - 2]) << 16) | (int(data[at + bsize - 1]) << 24)))
        at += bsize
    in_off = np.array([b[0] for b in blocks], np.int64)
    out_off = np.concatenate([[0], np.cumsum([b[1] for b in blocks])]).astype(np.int64)
    comp = pair.alloc(data.size + 64)
    comp[: data.size] = data
    outp = pair.alloc(int(out_off[-1]) + 64)
    for rep in range(3):
        t = time.time()
        e.inflate_blocks(comp, data.size, in_off, out_off, outp)
        dt = time.time() - t
    print("device, pinned host -> pinned host (both copies inside): %.1f ms = %.1f GB/s of output" % (dt * 1e3, out_off[-1] / dt / 1e9), flush=True)
    assert np.array_equal(outp[: int(out_off[-1])], got)
    pair.free_all()
    t = time.time()
    want = np.frombuffer(gzip.decompress(data.tobytes()), np.uint8)
    print("python gzip (one thread): %.2f s = %.2f GB/s" % (time.time() - t, want.size / (time.time() - t) / 1e9))
    assert np.array_equal(got, want)
    t = time.time()
    tb = io_native.read_bam_table(bam)
    dt = time.time() - t
    print("host library, whole-file decode (inflate + records, %d threads, %s): %.2f s = %.1f GB/s of output" % (
        io_native.default_threads(), io_native.inflate_backend(), dt, want.size / dt / 1e9))
finally:
    shutil.rmtree(d, ignore_errors=True)
