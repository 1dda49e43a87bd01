"""CRAM 3.0 decoder into unfazed_amd.model.Segment (what the reference gets from
`pysam.AlignmentFile(path, "rc", reference_filename=...)`, read_collector.py:372-373, unfazed.py:97-126).

Written from the published CRAM 3.0 format description (containers / slices / blocks, ITF8 / LTF8 integers, the core
bit stream with HUFFMAN / BETA / GAMMA / SUBEXP codes, EXTERNAL / BYTE_ARRAY_LEN / BYTE_ARRAY_STOP over external
blocks, raw / gzip / bzip2 / lzma / rANS 4x8 block compression, read features against the reference sequence, mate
reconstruction inside a slice).  The image has no htslib and the reference ships no CRAM file, so this decoder is
pinned only by round trips through the test-side writer (tests/cramio.py) -- "unpinned" against htslib output,
DESIGN.md section 4 says so.  Not decoded, rejected loudly: CRAM 2.x, the 3.1 block codecs (rANS Nx16, adaptive
arithmetic, fqzcomp, name tokeniser), GOLOMB / GOLOMB_RICE codes.

The records come out exactly as io_bam.read_bam gives them (flag, tid, pos, mapq, CIGAR, mate tid / pos, tlen, name,
bases, qualities, SA tag present), so everything downstream (model.ReadsTable, staging, the device path) is shared.
Region decode goes through the .crai index: only the slices the intervals (and the mates of the records they
return) touch are read and decoded.
"""
from __future__ import annotations

import bisect
import bz2
import gzip
import hashlib
import lzma
import os
import struct
import zlib
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

from .model import Segment

FUNMAP, FMUNMAP, FREVERSE, FMREVERSE, FREAD1 = 4, 8, 16, 32, 64
CF_QUAL_ARRAY, CF_DETACHED, CF_MATE_DOWNSTREAM, CF_NO_SEQ = 1, 2, 4, 8
_OP = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4, "H": 5, "P": 6}
_INT_SERIES = ("BF", "CF", "RI", "RL", "AP", "RG", "MF", "NS", "NP", "TS", "NF", "TL", "FN", "FP", "DL", "RS", "PD", "HC", "MQ")
_BYTE_SERIES = ("FC", "BA", "QS", "BS")
_ARRAY_SERIES = ("RN", "IN", "SC", "BB", "QQ")


class CramError(ValueError):
    pass


# ------------------------------------------------------------------ integers / cursors
class _In:
    __slots__ = ("b", "p")

    def __init__(self, b, p=0):
        self.b, self.p = b, p

    def u8(self) -> int:
        v = self.b[self.p]
        self.p += 1
        return v

    def i32(self) -> int:
        (v,) = struct.unpack_from("<i", self.b, self.p)
        self.p += 4
        return v

    def take(self, n: int) -> bytes:
        v = bytes(self.b[self.p: self.p + n])
        if len(v) != n:
            raise CramError("truncated CRAM data")
        self.p += n
        return v

    def until(self, stop: int) -> bytes:
        e = self.b.index(stop, self.p)
        v = bytes(self.b[self.p: e])
        self.p = e + 1
        return v

    def itf8(self) -> int:
        b, p = self.b, self.p
        b0 = b[p]
        if b0 < 0x80:
            self.p = p + 1
            return b0
        if b0 < 0xC0:
            self.p = p + 2
            return ((b0 & 0x3F) << 8) | b[p + 1]
        if b0 < 0xE0:
            self.p = p + 3
            return ((b0 & 0x1F) << 16) | (b[p + 1] << 8) | b[p + 2]
        if b0 < 0xF0:
            self.p = p + 4
            return ((b0 & 0x0F) << 24) | (b[p + 1] << 16) | (b[p + 2] << 8) | b[p + 3]
        self.p = p + 5
        v = ((b0 & 0x0F) << 28) | (b[p + 1] << 20) | (b[p + 2] << 12) | (b[p + 3] << 4) | (b[p + 4] & 0x0F)
        return v - (1 << 32) if v >= (1 << 31) else v

    def ltf8(self) -> int:
        b0 = self.u8()
        extra, lead = 0, 0x80
        while extra < 8 and (b0 & lead):
            extra += 1
            lead >>= 1
        v = b0 & (lead - 1) if extra < 8 else 0
        for _ in range(extra):
            v = (v << 8) | self.u8()
        return v - (1 << 64) if v >= (1 << 63) else v

    def itf8_array(self) -> List[int]:
        return [self.itf8() for _ in range(self.itf8())]


class _Bits:
    """the core block: bits are taken most significant first"""
    __slots__ = ("d", "p", "n")

    def __init__(self, d: bytes):
        self.d, self.p, self.n = d, 0, 8 * len(d)

    def bits(self, k: int) -> int:
        if k == 0:
            return 0
        p = self.p
        e = p + k
        if e > self.n:
            raise CramError("core block exhausted")
        b1 = (e + 7) >> 3
        v = int.from_bytes(self.d[p >> 3: b1], "big") >> (b1 * 8 - e)
        self.p = e
        return v & ((1 << k) - 1)


# ------------------------------------------------------------------ rANS 4x8
def _rans_freqs(inp: _In):
    """one frequency table: symbols in ascending order, runs of consecutive symbols run-length coded"""
    freq = [0] * 256
    rle = 0
    j = inp.u8()
    while True:
        f = inp.u8()
        if f >= 128:
            f = ((f & 127) << 8) | inp.u8()
        freq[j] = f
        if rle == 0 and inp.b[inp.p] == j + 1:
            j = inp.u8()
            rle = inp.u8()
        elif rle:
            rle -= 1
            j += 1
        else:
            j = inp.u8()
        if j == 0:
            break
    cum, look, x = [0] * 256, bytearray(4096), 0
    for s in range(256):
        f = freq[s]
        if f:
            cum[s] = x
            if x + f > 4096:
                raise CramError("rANS frequencies exceed 4096")
            look[x: x + f] = bytes([s]) * f
            x += f
    return freq, cum, look


def rans4x8_decode(data: bytes) -> bytes:
    inp = _In(data)
    order = inp.u8()
    (n_in, n_out) = struct.unpack_from("<II", data, 1)
    inp.p = 9
    if n_in != len(data) - 9:
        raise CramError("rANS block size mismatch")
    if n_out == 0:
        return b""
    out = bytearray(n_out)
    b = data
    if order == 0:
        freq, cum, look = _rans_freqs(inp)
        st = list(struct.unpack_from("<4I", b, inp.p))
        p = inp.p + 16
        nb = len(b)
        for i in range(n_out):
            k = i & 3
            x = st[k]
            m = x & 0xFFF
            s = look[m]
            out[i] = s
            x = freq[s] * (x >> 12) + m - cum[s]
            while x < (1 << 23) and p < nb:
                x = (x << 8) | b[p]
                p += 1
            st[k] = x
        return bytes(out)
    if order != 1:
        raise CramError("unknown rANS order %d" % order)
    tabs: Dict[int, tuple] = {}
    rle = 0
    i = inp.u8()
    while True:
        tabs[i] = _rans_freqs(inp)
        if rle == 0 and inp.b[inp.p] == i + 1:
            i = inp.u8()
            rle = inp.u8()
        elif rle:
            rle -= 1
            i += 1
        else:
            i = inp.u8()
        if i == 0:
            break
    st = list(struct.unpack_from("<4I", b, inp.p))
    p = inp.p + 16
    nb = len(b)
    q = n_out >> 2
    idx = [0, q, 2 * q, 3 * q]
    last = [0, 0, 0, 0]
    for _ in range(q):
        for k in range(4):
            freq, cum, look = tabs[last[k]]
            x = st[k]
            m = x & 0xFFF
            s = look[m]
            out[idx[k]] = s
            idx[k] += 1
            x = freq[s] * (x >> 12) + m - cum[s]
            while x < (1 << 23) and p < nb:
                x = (x << 8) | b[p]
                p += 1
            st[k] = x
            last[k] = s
    j = idx[3]
    while j < n_out:  # the remainder rides on the fourth state
        freq, cum, look = tabs[last[3]]
        x = st[3]
        m = x & 0xFFF
        s = look[m]
        out[j] = s
        j += 1
        x = freq[s] * (x >> 12) + m - cum[s]
        while x < (1 << 23) and p < nb:
            x = (x << 8) | b[p]
            p += 1
        st[3] = x
        last[3] = s
    return bytes(out)


_rans = rans4x8_decode


def _use_native_rans():
    """the io library's rANS decoder when it is built (same result, two orders of magnitude faster)"""
    global _rans
    try:
        from . import io_native
        f = io_native.rans4x8_decode
    except Exception:  # library not built: the Python loop above
        return
    _rans = f


# ------------------------------------------------------------------ blocks
class Block:
    __slots__ = ("method", "ctype", "cid", "data")

    def __init__(self, method, ctype, cid, data):
        self.method, self.ctype, self.cid, self.data = method, ctype, cid, data


_METHODS_31 = {5: "rANS Nx16", 6: "adaptive arithmetic", 7: "fqzcomp", 8: "name tokeniser"}


def _read_block(inp: _In) -> Block:
    p0 = inp.p
    method, ctype = inp.u8(), inp.u8()
    cid, csize, rsize = inp.itf8(), inp.itf8(), inp.itf8()
    raw = inp.take(csize)
    (crc,) = struct.unpack("<I", inp.take(4))
    if zlib.crc32(bytes(inp.b[p0: inp.p - 4])) & 0xFFFFFFFF != crc:
        raise CramError("CRAM block checksum mismatch")
    if method == 0:
        data = raw
    elif method == 1:
        data = zlib.decompress(raw, 31)
    elif method == 2:
        data = bz2.decompress(raw)
    elif method == 3:
        data = lzma.decompress(raw)
    elif method == 4:
        data = _rans(raw)
    elif method in _METHODS_31:
        raise CramError("CRAM 3.1 block codec (%s) is not decoded by this build: rewrite the file as CRAM 3.0" % _METHODS_31[method])
    else:
        raise CramError("unknown CRAM block compression method %d" % method)
    if len(data) != rsize:
        raise CramError("CRAM block inflates to %d bytes, header says %d" % (len(data), rsize))
    return Block(method, ctype, cid, data)


# ------------------------------------------------------------------ encodings
def _read_encoding(inp: _In):
    kind = inp.itf8()
    n = inp.itf8()
    sub = _In(inp.b, inp.p)
    inp.p += n
    if kind == 0:
        return (0,)
    if kind == 1:
        return (1, sub.itf8())
    if kind in (2, 8):
        return (kind, sub.itf8(), sub.itf8())
    if kind == 3:
        syms = sub.itf8_array()
        lens = sub.itf8_array()
        return (3, syms, lens)
    if kind == 4:
        return (4, _read_encoding(sub), _read_encoding(sub))
    if kind == 5:
        stop = sub.u8()
        return (5, stop, sub.itf8())
    if kind == 6:
        return (6, sub.itf8(), sub.itf8())
    if kind == 7:
        return (7, sub.itf8(), sub.itf8())
    if kind == 9:
        return (9, sub.itf8())
    raise CramError("unknown CRAM encoding id %d" % kind)


class CompressionHeader:
    def __init__(self, data: bytes):
        inp = _In(data)
        self.rn, self.ap_delta, self.rr = True, True, True
        self.sm = bytes([0x1B] * 5)
        self.td: List[List[bytes]] = [[]]
        inp.itf8()  # size in bytes
        for _ in range(inp.itf8()):
            key = inp.take(2)
            if key == b"RN":
                self.rn = bool(inp.u8())
            elif key == b"AP":
                self.ap_delta = bool(inp.u8())
            elif key == b"RR":
                self.rr = bool(inp.u8())
            elif key == b"SM":
                self.sm = inp.take(5)
            elif key == b"TD":
                blob = inp.take(inp.itf8())
                lines = blob.split(b"\0")
                if lines and lines[-1] == b"":
                    lines.pop()
                self.td = [[ln[i: i + 3] for i in range(0, len(ln), 3)] for ln in lines] or [[]]
            else:
                raise CramError("unknown preservation map key %r" % key)
        self.series: Dict[str, tuple] = {}
        inp.itf8()
        for _ in range(inp.itf8()):
            key = inp.take(2).decode("latin1")
            self.series[key] = _read_encoding(inp)
        self.tags: Dict[int, tuple] = {}
        inp.itf8()
        for _ in range(inp.itf8()):
            key = inp.itf8()
            self.tags[key] = _read_encoding(inp)
        # substitution matrix: for every reference base (A C G T N) the four other bases, each with its two-bit code
        self.subst = {}
        for r, ref in enumerate(b"ACGTN"):
            others = [x for x in b"ACGTN" if x != ref]
            row = [0] * 4
            for k, alt in enumerate(others):
                row[(self.sm[r] >> (6 - 2 * k)) & 3] = alt
            self.subst[ref] = row


class _SliceStreams:
    """the core bit stream and one cursor per external block of a slice, and decoders built on them"""

    def __init__(self, core: bytes, ext: Dict[int, bytes]):
        self.core = _Bits(core)
        self.ext = {cid: _In(d) for cid, d in ext.items()}

    def _cursor(self, cid: int) -> _In:
        c = self.ext.get(cid)
        if c is None:  # a series that is declared but never used in this slice has no block: reading it runs off the end
            c = self.ext[cid] = _In(b"")
        return c

    def _huffman(self, syms, lens):
        if len(syms) != len(lens) or not syms:
            raise CramError("bad HUFFMAN code")
        if len(syms) == 1 and lens[0] == 0:
            v = syms[0]
            return lambda: v
        order = sorted(zip(lens, syms))
        table, code, prev = {}, 0, order[0][0]
        for ln, s in order:
            code <<= ln - prev
            table[(ln, code)] = s
            code += 1
            prev = ln
        steps = sorted(set(lens))
        core = self.core

        def dec():
            code, have = 0, 0
            for ln in steps:
                code = (code << (ln - have)) | core.bits(ln - have)
                have = ln
                s = table.get((ln, code))
                if s is not None:
                    return s
            raise CramError("bad HUFFMAN code word")
        return dec

    def value(self, enc, as_byte: bool):
        """decoder of one integer (ITF8 in external blocks) or one byte"""
        kind = enc[0]
        if kind == 1:
            c = self._cursor(enc[1])
            return c.u8 if as_byte else c.itf8
        if kind == 3:
            return self._huffman(enc[1], enc[2])
        core = self.core
        if kind == 6:
            off, nb = enc[1], enc[2]
            return lambda: core.bits(nb) - off
        if kind == 9:
            off = enc[1]

            def gamma():
                n = 0
                while core.bits(1) == 0:
                    n += 1
                return ((1 << n) | core.bits(n)) - off
            return gamma
        if kind == 7:
            off, k = enc[1], enc[2]

            def subexp():
                u = 0
                while core.bits(1) == 1:
                    u += 1
                if u == 0:
                    return core.bits(k) - off
                b = u + k - 1
                return ((1 << b) | core.bits(b)) - off
            return subexp
        if kind in (2, 8):
            raise CramError("GOLOMB / GOLOMB_RICE codes are not decoded by this build")
        if kind == 0:
            def null():
                raise CramError("data series with a NULL encoding is read")
            return null
        raise CramError("encoding %d cannot code a single value" % kind)

    def nbytes(self, enc):
        """decoder of n values of a byte series at once"""
        if enc[0] == 1:
            return self._cursor(enc[1]).take
        one = self.value(enc, True)
        return lambda n: bytes(one() & 0xFF for _ in range(n))

    def array(self, enc):
        """decoder of one byte array"""
        kind = enc[0]
        if kind == 4:
            ln = self.value(enc[1], False)
            val = self.nbytes(enc[2])
            return lambda: val(ln())
        if kind == 5:
            c = self._cursor(enc[2])
            stop = enc[1]
            return lambda: c.until(stop)
        if kind == 0:
            def null():
                raise CramError("data series with a NULL encoding is read")
            return null
        raise CramError("encoding %d cannot code a byte array" % kind)


# ------------------------------------------------------------------ reference sequence
class Fasta:
    """indexed FASTA (NAME.fai next to it, else the file is scanned once)"""

    def __init__(self, path: str):
        self.path = path
        self.index: Dict[str, Tuple[int, int, int, int]] = {}
        fai = path + ".fai"
        if os.path.isfile(fai):
            with open(fai) as fh:
                for ln in fh:
                    f = ln.rstrip("\n").split("\t")
                    if len(f) >= 5:
                        self.index[f[0]] = (int(f[1]), int(f[2]), int(f[3]), int(f[4]))
        else:
            self._scan()
        self.fh = open(path, "rb")

    def _scan(self):
        opener = gzip.open if self.path.endswith(".gz") else open
        if opener is gzip.open:
            raise CramError("a compressed reference FASTA needs its .fai (and must be BGZF with a .gzi): use an uncompressed FASTA")
        with open(self.path, "rb") as fh:
            name, off, length, lb, lw = None, 0, 0, 0, 0
            pos = 0
            for ln in fh:
                if ln.startswith(b">"):
                    if name is not None:
                        self.index[name] = (length, off, lb, lw)
                    name = ln[1:].split()[0].decode()
                    off, length, lb, lw = pos + len(ln), 0, 0, 0
                else:
                    body = len(ln.rstrip(b"\r\n"))
                    if lb == 0:
                        lb, lw = body, len(ln)
                    length += body
                pos += len(ln)
            if name is not None:
                self.index[name] = (length, off, lb, lw)

    def fetch(self, name: str, start: int, end: int) -> bytes:
        """bases [start, end) (0-based), upper case; clipped to the contig"""
        if name not in self.index:
            raise CramError("contig %s is not in the reference FASTA %s" % (name, self.path))
        length, off, lb, lw = self.index[name]
        start, end = max(0, start), min(end, length)
        if end <= start or lb <= 0:
            return b""
        a = off + (start // lb) * lw + start % lb
        b = off + ((end - 1) // lb) * lw + (end - 1) % lb + 1
        self.fh.seek(a)
        return self.fh.read(b - a).replace(b"\n", b"").replace(b"\r", b"").upper()

    def close(self):
        self.fh.close()


class _RefWindow:
    """reference bases of the contig a slice (or a record of a multi-reference slice) sits on"""

    def __init__(self, fasta: Optional[Fasta], names: Sequence[str]):
        self.fasta, self.names = fasta, names
        self.tid, self.lo, self.seq = -9, 0, b""

    def set_embedded(self, tid: int, start0: int, seq: bytes):
        self.tid, self.lo, self.seq = tid, start0, seq.upper()

    def get(self, tid: int, a: int, b: int) -> bytes:
        """bases [a, b) 0-based; short (or empty) past the end of the contig"""
        if b <= a:
            return b""
        if tid != self.tid or a < self.lo or b > self.lo + len(self.seq):
            if self.fasta is None:
                raise CramError("this CRAM needs its reference FASTA (-r / --reference)")
            lo = max(0, a - 1000)
            seq = self.fasta.fetch(self.names[tid], lo, max(b, a + 100000))
            self.tid, self.lo, self.seq = tid, lo, seq
        return self.seq[a - self.lo: b - self.lo]


# ------------------------------------------------------------------ file structure
class _Container:
    __slots__ = ("offset", "length", "ref", "start", "span", "n_records", "counter", "n_blocks", "landmarks", "body_off")


class _SliceHeader:
    __slots__ = ("ref", "start", "span", "n_records", "counter", "n_blocks", "ids", "embedded", "md5")

    def __init__(self, data: bytes):
        inp = _In(data)
        self.ref, self.start, self.span, self.n_records = inp.itf8(), inp.itf8(), inp.itf8(), inp.itf8()
        self.counter = inp.ltf8()
        self.n_blocks = inp.itf8()
        self.ids = inp.itf8_array()
        self.embedded = inp.itf8()
        self.md5 = inp.take(16)


class CramFile:
    def __init__(self, path: str, reference: Optional[str] = None):
        self.path = path
        self.fh = open(path, "rb")
        head = self.fh.read(26)
        if head[:4] != b"CRAM":
            raise CramError("%s is not a CRAM file" % path)
        self.major, self.minor = head[4], head[5]
        if self.major != 3:
            raise CramError("%s is CRAM %d.%d: only CRAM 3.0 is decoded by this build" % (path, self.major, self.minor))
        _use_native_rans()
        self.slices_decoded = 0
        self.bytes_read = 0
        first = self._container_at(26)
        body = self._body(first)
        blk = _read_block(_In(body))
        (l_text,) = struct.unpack_from("<i", blk.data, 0)
        self.text = blk.data[4: 4 + l_text].decode("latin1")
        self.contigs: List[str] = []
        self.contig_len: List[int] = []
        for ln in self.text.split("\n"):
            if ln.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in ln.split("\t")[1:] if ":" in x)
                self.contigs.append(f["SN"])
                self.contig_len.append(int(f.get("LN", "0")))
        self.data_off = first.body_off + first.length
        self.fasta = Fasta(reference) if reference else None

    def close(self):
        self.fh.close()
        if self.fasta:
            self.fasta.close()

    def _container_at(self, off: int) -> Optional[_Container]:
        self.fh.seek(off)
        raw = self.fh.read(64 * 1024)
        if len(raw) < 4:
            return None
        inp = _In(raw)
        c = _Container()
        c.offset = off
        c.length = inp.i32()
        c.ref, c.start, c.span, c.n_records = inp.itf8(), inp.itf8(), inp.itf8(), inp.itf8()
        c.counter, _bases = inp.ltf8(), inp.ltf8()
        c.n_blocks = inp.itf8()
        c.landmarks = inp.itf8_array()
        (crc,) = struct.unpack("<I", inp.take(4))
        if zlib.crc32(raw[: inp.p - 4]) & 0xFFFFFFFF != crc:
            raise CramError("CRAM container header checksum mismatch at offset %d" % off)
        c.body_off = off + inp.p
        return c

    def _body(self, c: _Container) -> bytes:
        self.fh.seek(c.body_off)
        b = self.fh.read(c.length)
        if len(b) != c.length:
            raise CramError("truncated CRAM container at offset %d" % c.offset)
        self.bytes_read += len(b)
        return b

    def containers(self) -> Iterable[_Container]:
        off = self.data_off
        while True:
            c = self._container_at(off)
            if c is None:
                return
            if c.n_records > 0 or not (c.ref == -1 and c.start == 4542278):  # (the end-of-file container)
                yield c
            off = c.body_off + c.length

    # ---- slices
    def decode_container(self, c: _Container, only_slices: Optional[Sequence[int]] = None, as_bam: bool = False):
        """-> [(slice offset within the container, records)]; as_bam: the records of a slice as uncompressed BAM records
        (bytes) from the native record layer (uz_cram_slice_to_bam) instead of Segment objects"""
        if c.n_records == 0:
            return []
        body = self._body(c)
        inp = _In(body)
        ch_block = _read_block(inp)
        if ch_block.ctype != 1:
            raise CramError("container does not start with a compression header")
        ch = CompressionHeader(ch_block.data)
        out = []
        for lm in c.landmarks:
            if only_slices is not None and lm not in only_slices:
                continue
            out.append((lm, self._decode_slice(ch, _In(body, lm), ch_block.data if as_bam else None)))
        return out

    def _decode_slice(self, ch: CompressionHeader, inp: _In, ch_raw: Optional[bytes] = None):
        hb = _read_block(inp)
        if hb.ctype != 2:
            raise CramError("slice does not start with a slice header block")
        sh = _SliceHeader(hb.data)
        core, ext = b"", {}
        for _ in range(sh.n_blocks):
            b = _read_block(inp)
            if b.ctype == 5:
                core = b.data
            elif b.ctype == 4:
                ext[b.cid] = b.data
        self.slices_decoded += 1
        if ch_raw is not None and sh.ref != -2:
            from . import io_native
            ref, ref0 = None, 0
            if sh.embedded >= 0:
                if sh.embedded not in ext:
                    raise CramError("slice names a missing embedded reference block")
                ref, ref0 = bytes(ext[sh.embedded]).upper(), sh.start - 1
            elif sh.ref >= 0 and self.fasta is not None:
                ref0 = max(0, sh.start - 1)
                ref = self.fasta.fetch(self.contigs[sh.ref], ref0, sh.start - 1 + sh.span + 1000)
                _check_slice_md5(self, sh, ref[sh.start - 1 - ref0: sh.start - 1 - ref0 + sh.span], ch)
            elif sh.ref >= 0 and ch.rr:
                raise CramError("this CRAM needs its reference FASTA (-r / --reference)")
            try:
                return io_native.cram_slice_to_bam(ch_raw, sh.ref, sh.start, sh.span, sh.n_records, sh.counter, core, ext, ref, ref0)
            except io_native.IoError as e:
                raise CramError("%s: %s" % (self.path, e))
        try:
            recs = _decode_records(ch, sh, _SliceStreams(core, ext), self)
        except IndexError:
            raise CramError("a data series of %s runs past the end of its block" % self.path)
        return b"".join(bam_record(s) for s in recs) if ch_raw is not None else recs  # (multi-reference slice: packed here)


def _check_slice_md5(cf: "CramFile", sh: "_SliceHeader", seq: bytes, ch: "CompressionHeader") -> None:
    if ch.rr and any(sh.md5) and hashlib.md5(seq).digest() != sh.md5:
        raise CramError("the reference FASTA does not match the one %s was written against (contig %s, %d-%d)"
                        % (cf.path, cf.contigs[sh.ref], sh.start, sh.start + sh.span - 1))


_SEQ16 = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def bam_record(s: Segment) -> bytes:
    """one Segment as an uncompressed BAM record (block_size word included; an SA:Z:* tag when the record carries one)"""
    name = s.qname.encode("latin1") + b"\0"
    l_seq = len(s.seq) if s.seq else 0
    sq = bytearray((l_seq + 1) // 2)
    for i, ch in enumerate(s.seq or ""):
        sq[i >> 1] |= _SEQ16.get(ch.upper(), 15) << (4 if i % 2 == 0 else 0)
    qual = bytes(s.qual[:l_seq]) if (s.qual is not None and l_seq) else b"\xff" * l_seq
    body = struct.pack("<iiBBHHHiiii", s.tid, s.pos, len(name), s.mapq & 0xFF, 4680, len(s.cigar), s.flag & 0xFFFF, l_seq, s.mtid, s.mpos,
                       s.tlen) + name + b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in s.cigar) + bytes(sq) + qual + \
        (b"SAZ*\0" if s.has_sa else b"")
    return struct.pack("<i", len(body)) + body


def bam_stream_header(cf: "CramFile") -> bytes:
    text = cf.text.encode("latin1")
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(cf.contigs)))
    for name, ln in zip(cf.contigs, cf.contig_len):
        nm = name.encode("latin1") + b"\0"
        out += struct.pack("<i", len(nm)) + nm + struct.pack("<i", ln)
    return bytes(out)


def _decode_records(ch: CompressionHeader, sh: _SliceHeader, st: _SliceStreams, cf: CramFile) -> List[Segment]:
    S = ch.series

    def need(name):
        if name not in S:
            def missing():
                raise CramError("data series %s is read but has no encoding" % name)
            return missing
        if name in _ARRAY_SERIES:
            return st.array(S[name])
        return st.value(S[name], name in _BYTE_SERIES)
    d = {k: need(k) for k in _INT_SERIES + _BYTE_SERIES + _ARRAY_SERIES}
    qs_n = st.nbytes(S["QS"]) if "QS" in S else None
    ba_n = st.nbytes(S["BA"]) if "BA" in S else None
    tag_dec = {}
    ref = _RefWindow(cf.fasta, cf.contigs)
    if sh.embedded >= 0:
        if sh.embedded not in st.ext:
            raise CramError("slice names a missing embedded reference block")
        ref.set_embedded(sh.ref, sh.start - 1, st.ext[sh.embedded].b)
    elif sh.ref >= 0 and ch.rr and cf.fasta is not None and any(sh.md5):
        _check_slice_md5(cf, sh, ref.get(sh.ref, sh.start - 1, sh.start - 1 + sh.span), ch)
    n = sh.n_records
    recs = [None] * n  # [flag, cf, tid, pos1, aend1, mapq, cigar, seq, qual, name, mate_line, mtid, mpos1, tlen, has_sa]
    last_pos = sh.start
    multi = sh.ref == -2
    subst = ch.subst
    for i in range(n):
        bf = d["BF"]()
        cfl = d["CF"]()
        tid = d["RI"]() if multi else sh.ref
        rl = d["RL"]()
        ap = d["AP"]()
        if ch.ap_delta:
            ap += last_pos
            last_pos = ap
        d["RG"]()
        name = d["RN"]() if ch.rn else None
        mate_line, mtid, mpos, tlen = -1, -1, 0, 0
        if cfl & CF_DETACHED:
            mf = d["MF"]()
            if not ch.rn:
                name = d["RN"]()
            mtid, mpos, tlen = d["NS"](), d["NP"](), d["TS"]()
            if mf & 1:
                bf |= FMREVERSE
            if mf & 2:
                bf |= FMUNMAP
        elif cfl & CF_MATE_DOWNSTREAM:
            mate_line = i + d["NF"]() + 1
            tlen = None
        tl = d["TL"]()
        if not 0 <= tl < len(ch.td):
            raise CramError("tag line %d is outside the tag dictionary" % tl)
        has_sa = False
        for ent in ch.td[tl]:
            key = (ent[0] << 16) | (ent[1] << 8) | ent[2]
            f = tag_dec.get(key)
            if f is None:
                if key not in ch.tags:
                    raise CramError("tag %r has no encoding" % ent)
                f = tag_dec[key] = st.array(ch.tags[key])
            f()
            has_sa |= ent[:2] == b"SA"
        cigar: List[List[int]] = []

        def op(code, ln):
            if ln <= 0:
                return
            if cigar and cigar[-1][0] == code:
                cigar[-1][1] += ln
            else:
                cigar.append([code, ln])
        qual = None
        if not bf & FUNMAP:
            seq = bytearray(rl)
            qpatch = []
            rpos, spos = ap - 1, 0  # next reference base (0-based), next read base (0-based)
            prev = 0
            for _ in range(d["FN"]()):
                code = d["FC"]()
                prev += d["FP"]()
                fpos = prev - 1
                if fpos > spos:
                    ln = fpos - spos
                    if not cfl & CF_NO_SEQ:
                        m = ref.get(tid, rpos, rpos + ln)
                        seq[spos: spos + len(m)] = m
                        if len(m) < ln:
                            seq[spos + len(m): spos + ln] = b"N" * (ln - len(m))
                    op(0, ln)
                    rpos += ln
                    spos = fpos
                if code == 88:  # X: substitution
                    r = ref.get(tid, rpos, rpos + 1) or b"N"
                    seq[spos] = subst.get(r[0], subst[78])[d["BS"]() & 3]
                    op(0, 1)
                    rpos += 1
                    spos += 1
                elif code == 66:  # B: base and quality
                    seq[spos] = d["BA"]()
                    qpatch.append((spos, bytes([d["QS"]() & 0xFF])))
                    op(0, 1)
                    rpos += 1
                    spos += 1
                elif code == 98:  # b: stretch of bases
                    bb = d["BB"]()
                    seq[spos: spos + len(bb)] = bb
                    op(0, len(bb))
                    rpos += len(bb)
                    spos += len(bb)
                elif code == 73:  # I
                    ins = d["IN"]()
                    seq[spos: spos + len(ins)] = ins
                    op(1, len(ins))
                    spos += len(ins)
                elif code == 105:  # i: one inserted base
                    seq[spos] = d["BA"]()
                    op(1, 1)
                    spos += 1
                elif code == 83:  # S
                    sc = d["SC"]()
                    seq[spos: spos + len(sc)] = sc
                    op(4, len(sc))
                    spos += len(sc)
                elif code == 68:  # D
                    ln = d["DL"]()
                    op(2, ln)
                    rpos += ln
                elif code == 78:  # N
                    ln = d["RS"]()
                    op(3, ln)
                    rpos += ln
                elif code == 72:  # H
                    op(5, d["HC"]())
                elif code == 80:  # P
                    op(6, d["PD"]())
                elif code == 81:  # Q: one quality
                    qpatch.append((fpos, bytes([d["QS"]() & 0xFF])))
                elif code == 113:  # q: stretch of qualities
                    qpatch.append((fpos, d["QQ"]()))
                else:
                    raise CramError("unknown read feature %r" % chr(code))
            if spos < rl:
                ln = rl - spos
                if not cfl & CF_NO_SEQ:
                    m = ref.get(tid, rpos, rpos + ln)
                    seq[spos: spos + len(m)] = m
                    if len(m) < ln:
                        seq[spos + len(m): rl] = b"N" * (ln - len(m))
                op(0, ln)
                rpos += ln
            if len(seq) != rl:
                raise CramError("read features overrun the read length")
            mapq = d["MQ"]()
            if cfl & CF_QUAL_ARRAY:
                qual = bytearray(qs_n(rl))  # (the whole array: it overrides what features carried)
            elif qpatch:
                qual = bytearray(b"\xff" * rl)
                for at, q in qpatch:
                    qual[at: at + len(q)] = q
            aend = max(rpos, ap)  # 1-based inclusive end
            seq_s = "" if cfl & CF_NO_SEQ else seq.decode("latin1")
        else:
            seq_s = "" if cfl & CF_NO_SEQ else ba_n(rl).decode("latin1")
            if cfl & CF_QUAL_ARRAY:
                qual = bytearray(qs_n(rl))
            mapq = 0
            aend = ap
        if qual is not None and (len(qual) == 0 or all(q == 0xFF for q in qual)):
            qual = None
        recs[i] = [bf, cfl, tid, ap, aend, mapq, [(c, ln) for c, ln in cigar], seq_s, None if qual is None else list(qual),
                   name, mate_line, mtid, mpos, tlen, has_sa]

    # mates inside the slice (chains through "records to the next fragment"); template length over the chain
    for i in range(n):
        r = recs[i]
        if r[10] < 0 or r[13] is not None:
            continue
        if r[10] >= n:
            raise CramError("mate link leaves the slice")
        chain = [i]
        j = i
        while recs[j][10] >= 0 and len(chain) <= n:
            nxt = recs[j][10]
            if nxt <= j or nxt >= n:
                raise CramError("bad mate link")
            chain.append(nxt)
            j = nxt
        left = min(recs[k][3] for k in chain)
        right = max(recs[k][4] for k in chain)
        left_cnt = sum(1 for k in chain if recs[k][3] == left)
        same_ref = all(recs[k][2] == r[2] for k in chain)
        for pos_in_chain, k in enumerate(chain):
            rk = recs[k]
            mate = recs[chain[(pos_in_chain + 1) % len(chain)]]
            if same_ref:
                t = right - left + 1
                rk[13] = t if (rk[3] == left and (left_cnt == 1 or rk[0] & FREAD1)) else -t
            else:
                rk[13] = 0
            rk[11], rk[12] = mate[2], mate[3]
            if mate[0] & FUNMAP:
                rk[0] |= FMUNMAP
                rk[13] = 0
            if rk[0] & FUNMAP:
                rk[13] = 0
            if mate[0] & FREVERSE:
                rk[0] |= FMREVERSE
            if rk[9] is None:  # (names not kept: the records of a chain share the name generated for its first record)
                if recs[chain[0]][9] is None:
                    recs[chain[0]][9] = b"cram:%d" % (sh.counter + chain[0])
                rk[9] = recs[chain[0]][9]
            rk[10] = -2  # resolved
    out = []
    for i, r in enumerate(recs):
        name = r[9]
        if name is None:
            name = r[9] = b"cram:%d" % (sh.counter + i)
        tlen = r[13] if r[13] is not None else 0
        out.append(Segment(name.decode("latin1"), r[0] & 0xFFFF, r[2], r[3] - 1, r[5], r[6], r[11], r[12] - 1, tlen, r[7], r[8], r[14]))
    return out


# ------------------------------------------------------------------ whole-file and region decode
def read_cram(path: str, reference: Optional[str] = None, max_records: Optional[int] = None) -> Tuple[List[str], List[Segment]]:
    """-> (contig names, records in file order); max_records: stop after the slice that reaches it (head of the file)"""
    cf = CramFile(path, reference)
    try:
        segs: List[Segment] = []
        for c in cf.containers():
            for _, part in cf.decode_container(c):
                segs.extend(part)
            if max_records is not None and len(segs) >= max_records:
                break
        return cf.contigs, segs
    finally:
        cf.close()


def crai_path(path: str) -> Optional[str]:
    for cand in (path + ".crai", os.path.splitext(path)[0] + ".crai"):
        if os.path.isfile(cand):
            return cand
    return None


def read_crai(path: str) -> List[Tuple[int, int, int, int, int, int]]:
    """-> [(reference id, start (1-based), span, container offset, slice offset, slice size)]"""
    with gzip.open(path, "rt") as fh:
        return [tuple(int(x) for x in ln.split("\t")[:6]) for ln in fh if ln.strip()]


def _overlaps(seg: Segment, tid: int, lo: int, hi: int) -> bool:
    return seg.tid == tid and seg.pos < hi and seg.endpos > lo


class _Lite:
    """what the region selection looks at, read off an uncompressed BAM record (native record layer)"""
    __slots__ = ("raw", "tid", "pos", "endpos", "flag", "mtid", "mpos", "qname")


def _lite_records(buf: bytes) -> List[_Lite]:
    out = []
    off, n = 0, len(buf)
    unpack = struct.Struct("<iiiBBHHHiii").unpack_from
    while off + 4 <= n:
        bs, tid, pos, l_name, _mq, _bin, n_cig, flag, _l_seq, mtid, mpos = unpack(buf, off)
        r = _Lite()
        r.raw = buf[off: off + 4 + bs]
        r.tid, r.pos, r.flag, r.mtid, r.mpos = tid, pos, flag, mtid, mpos
        p = off + 36
        r.qname = buf[p: p + l_name - 1]
        p += l_name
        rl = 0
        if n_cig:
            for v in struct.unpack_from("<%dI" % n_cig, buf, p):
                if (v & 15) in (0, 2, 3, 7, 8):
                    rl += v >> 4
        r.endpos = pos + 1 if (flag & FUNMAP or not n_cig) else pos + (rl if rl > 0 else 1)
        out.append(r)
        off += 4 + bs
    return out


def read_cram_table(path: str, reference: Optional[str] = None, max_records: Optional[int] = None, threads: int = 0,
                    insert_size_max_sample: int = 1000000):
    """whole file (or its head: whole slices until max_records) -> model.ReadsTable through the native record layer and the
    BAM table builder (uz_cram_slice_to_bam, uz_bam_decode_memory)"""
    from . import io_native
    cf = CramFile(path, reference)
    try:
        parts, n = [bam_stream_header(cf)], 0
        for c in cf.containers():
            for _, part in cf.decode_container(c, as_bam=True):
                parts.append(part)
            n += c.n_records
            if max_records is not None and n >= max_records:
                break
        return io_native.read_bam_stream_table(b"".join(parts), threads=threads, insert_size_max_sample=insert_size_max_sample)
    finally:
        cf.close()


def read_cram_regions_table(path: str, reference: Optional[str], tid, lo, hi, crai: Optional[str] = None, stats: Optional[dict] = None,
                            threads: int = 0):
    """read_cram_regions through the native record layer -> model.ReadsTable"""
    from . import io_native
    header, kept = read_cram_regions(path, reference, tid, lo, hi, crai=crai, stats=stats, as_bam=True)
    return io_native.read_bam_stream_table(header + b"".join(r.raw for r in kept), threads=threads, insert_size_max_sample=0)


def read_cram_regions(path: str, reference: Optional[str], tid: Sequence[int], lo: Sequence[int], hi: Sequence[int],
                      crai: Optional[str] = None, stats: Optional[dict] = None, as_bam: bool = False):
    """What `fetch(contig, lo, hi)` returns for the intervals (start < hi and end > lo, 0-based half open) and, closed under
    it, what `mate()` returns for those records (same name at the mate position) -- the same contract as
    uz_bam_decode_regions (include/unfazed_io.h) -- in file order.  Only slices the index names for the intervals (and for
    the mate positions) are read."""
    crai = crai or crai_path(path)
    if crai is None:
        raise CramError("no .crai index next to %s" % path)
    index = read_crai(crai)
    cf = CramFile(path, reference)
    try:
        by_ref: Dict[int, List[tuple]] = {}
        for e in index:
            by_ref.setdefault(e[0], []).append(e)
        decoded: Dict[Tuple[int, int], List[Segment]] = {}

        def slices_for(points: Iterable[Tuple[int, int, int]]):
            want: Dict[int, set] = {}
            for t, a, b in points:
                for e in by_ref.get(t, ()):
                    if e[1] - 1 < b and e[1] - 1 + e[2] > a and (e[3], e[4]) not in decoded:
                        want.setdefault(e[3], set()).add(e[4])
            for coff in sorted(want):
                c = cf._container_at(coff)
                for lm, part in cf.decode_container(c, only_slices=want[coff], as_bam=as_bam):
                    decoded[(coff, lm)] = _lite_records(part) if as_bam else part

        ivs = sorted(set(zip((int(x) for x in tid), (int(x) for x in lo), (int(x) for x in hi))))
        slices_for(ivs)
        # union of the intervals per contig (disjoint, sorted): a record is fetched iff it overlaps the union
        union: Dict[int, Tuple[List[int], List[int]]] = {}
        for t, a, b in ivs:
            if b <= a:
                continue
            los, his = union.setdefault(t, ([], []))
            if his and a <= his[-1]:
                his[-1] = max(his[-1], b)
            else:
                los.append(a)
                his.append(b)

        def fetched(s: Segment) -> bool:
            u = union.get(s.tid)
            if u is None:
                return False
            k = bisect.bisect_left(u[1], s.pos + 1)  # first interval that ends after the record's start
            return k < len(u[0]) and u[0][k] < s.endpos
        keep: Dict[Tuple[int, int, int], Segment] = {}
        for key in sorted(decoded):
            for k, s in enumerate(decoded[key]):
                if s.tid >= 0 and fetched(s):
                    keep[key + (k,)] = s
        # mate(): the FIRST record in file order with the name that overlaps the mate position on the mate's contig and
        # carries the other read-of-pair flag; transitively (the read stage follows mate(mate(r))), generation by generation
        frontier = list(keep)
        for _ in range(64):
            if not frontier:
                break
            asks = []
            for key in frontier:
                s = keep[key]
                if s.flag & 1 and not s.flag & FMUNMAP and 0 <= s.mtid < len(cf.contigs):
                    asks.append((s.qname, s.mtid, s.mpos, (s.flag ^ 192) & 192))
            slices_for((t, p, p + 1) for _, t, p, _ in asks)
            by_name: Dict[str, List[Tuple[Tuple[int, int, int], Segment]]] = {}
            wanted = {a[0] for a in asks}
            for key in sorted(decoded):
                for k, s in enumerate(decoded[key]):
                    if s.qname in wanted:
                        by_name.setdefault(s.qname, []).append((key + (k,), s))
            frontier = []
            for name, t, p, want in asks:
                for key, s in by_name.get(name, ()):
                    if s.tid == t and s.pos < p + 1 and s.endpos > p and s.flag & want:
                        if key not in keep:
                            keep[key] = s
                            frontier.append(key)
                        break
        if stats is not None:
            stats.update(slices_decoded=cf.slices_decoded, bytes_read=cf.bytes_read, records_walked=sum(len(v) for v in decoded.values()),
                         records_kept=len(keep))
        return (bam_stream_header(cf) if as_bam else cf.contigs), [keep[k] for k in sorted(keep)]
    finally:
        cf.close()
