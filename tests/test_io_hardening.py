"""Broken input to the one-pass BAM stage (uz_bam_stage_*, csrc/io_stage.cpp): each case must come back as an IoError, never a crash.
The cases are the ones round 3's review listed: a record without a read name (l_read_name == 0), an index whose chunks point past the end
of a truncated file, a BGZF block that declares more than 64 KiB of inflated bytes, an extra subfield that overruns the extra field."""
import os
import struct
import zlib

import numpy as np
import pytest

import filesio
from unfazed_amd import io_bam, io_native


def _bgzf_block(data: bytes, isize=None, extra: bytes = None) -> bytes:
    comp = zlib.compressobj(6, zlib.DEFLATED, -15)
    cdata = comp.compress(data) + comp.flush()
    if extra is None:
        bsize = len(cdata) + 25
        extra = b"BC\x02\x00" + struct.pack("<H", bsize)
    else:  # caller-built extra field; BSIZE inside it must already be right
        pass
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", len(extra)) + extra + cdata
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) if isize is None else isize))


def _segs(n=40, start=1000):
    out = []
    for i in range(n):
        pos = start + 7 * i
        q = "r%04d" % i
        out.append(io_bam.Segment(q, 99, 0, pos, 60, [(0, 100)], 0, pos + 200, 300, "ACGT" * 25, [30] * 100, False))
    for i in range(n):
        pos = start + 7 * i + 200
        q = "r%04d" % i
        out.append(io_bam.Segment(q, 147, 0, pos, 60, [(0, 100)], 0, pos - 200, -300, "ACGT" * 25, [30] * 100, False))
    out.sort(key=lambda s: s.pos)
    return out


def _good_bam(path):
    io_bam.write_bam(path, [("1", 100000)], _segs())
    filesio.write_bai(path)


def _fetch():
    return (np.array([0], np.int32), np.array([1100], np.int32), np.array([1101], np.int32), np.array([1], np.uint16))


def _select(path):
    src = io_native.BamSource(path, threads=2)
    fc, flo, fhi, fex = _fetch()
    return src.select(fc, flo, fhi, 20, extra=fex)


def _rewrite(path, mutate_payload=None, block_kw=None):
    """inflate every block of `path`, let `mutate_payload(bytearray)` edit the stream, write it back as one-block-per-60000 BGZF (same cuts as the writer)"""
    raw = open(path, "rb").read()
    p, data = 0, bytearray()
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        data += zlib.decompress(raw[p + 12 + xlen: p + bsize - 8], -15)
        p += bsize
    if mutate_payload:
        mutate_payload(data)
    with open(path, "wb") as fh:
        for i in range(0, len(data), 60000):
            fh.write(_bgzf_block(bytes(data[i: i + 60000]), **(block_kw or {})))
        fh.write(_bgzf_block(b""))


def _first_record_at(data: bytes) -> int:
    l_text = struct.unpack_from("<i", data, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, p)[0]
        p += 4 + l_name + 4
    return p


def test_good_file_is_read(tmp_path):
    path = str(tmp_path / "ok.bam")
    _good_bam(path)
    got = _select(path)
    assert int(got.view.n_segs) > 0


def test_record_without_a_read_name_is_refused(tmp_path):
    path = str(tmp_path / "noname.bam")
    _good_bam(path)

    def mutate(d):
        at = _first_record_at(d)
        # walk to a record inside the fetch's reach and clear its l_read_name
        for _ in range(12):
            at += 4 + struct.unpack_from("<i", d, at)[0]
        d[at + 4 + 8] = 0
    _rewrite(path, mutate)
    with pytest.raises(io_native.IoError):
        _select(path)


def test_index_pointing_past_a_truncated_file_is_refused(tmp_path):
    path = str(tmp_path / "trunc.bam")
    # enough records for several BGZF blocks, so that chunks of the index begin in later blocks
    segs = sorted(_segs(n=1500, start=1000) + _segs(n=1500, start=5_000_000), key=lambda s: s.pos)
    io_bam.write_bam(path, [("1", 10_000_000)], segs)
    filesio.write_bai(path)
    raw = open(path, "rb").read()
    # keep the first block only (header + first records): the index still names the later ones
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1
    open(path, "wb").write(raw[:bsize])
    src = io_native.BamSource(path, threads=2, insert_size_max_sample=0)  # (the head of the file is intact: one record of it is read here)
    fc, flo, fhi, fex = (np.array([0], np.int32), np.array([5_000_500], np.int32), np.array([5_000_501], np.int32), np.array([1], np.uint16))
    with pytest.raises(io_native.IoError):
        src.select(fc, flo, fhi, 20, extra=fex)


def test_block_declaring_more_than_64k_is_refused(tmp_path):
    path = str(tmp_path / "isize.bam")
    _good_bam(path)
    _rewrite(path, None, {"isize": 0x7FFFFFF0})
    with pytest.raises(io_native.IoError):
        _select(path)


def test_extra_subfield_overrunning_the_extra_field_is_refused(tmp_path):
    path = str(tmp_path / "extra.bam")
    _good_bam(path)
    raw = bytearray(open(path, "rb").read())
    # first block: claim the BC subfield is 200 bytes long (it then runs past XLEN = 6)
    struct.pack_into("<H", raw, 12 + 2, 200)
    open(path, "wb").write(bytes(raw))
    with pytest.raises(io_native.IoError):
        _select(path)


def test_pool_survives_a_fork(tmp_path):
    """a forked child that never uses the pool leaves through the normal exit path (static destructors and all) without touching the
    parent's workers; one that does use it gets workers of its own"""
    import subprocess
    import sys
    path = str(tmp_path / "fork.bam")
    _good_bam(path)
    code = r"""
import os, sys
sys.path[:0] = [%r, %r]
import test_io_hardening as t
n0 = int(t._select(%r).view.n_segs)      # the parent's pool now has workers
for use in (False, True):
    pid = os.fork()
    if pid == 0:
        if use and int(t._select(%r).view.n_segs) != n0:
            os._exit(3)
        sys.exit(0)                       # the interpreter's normal way out
    _, st = os.waitpid(pid, 0)
    assert os.WIFEXITED(st) and os.WEXITSTATUS(st) == 0, (use, st)
print("ok", n0)
""" % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout, r.stderr[-2000:])


def test_cram_input_is_refused_with_the_way_out(tmp_path):
    """the reference opens CRAM through pysam (read_collector.py:372-373); this build has no decoder it could pin against htslib and says so"""
    from unfazed_amd import session
    p = tmp_path / "kid.cram"
    p.write_bytes(b"CRAM\x03\x00" + b"\x00" * 64)
    with pytest.raises(session.CramNotSupported, match="samtools view -b"):
        session.load_reads(str(p))
    lazy = session._LazyReads(1000)
    for call in (lambda: lazy.indexed(str(p)), lambda: lazy.header(str(p)), lambda: lazy.regions(str(p), [0], [1], [2])):
        with pytest.raises(session.CramNotSupported):
            call()
