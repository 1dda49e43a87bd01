"""BGZF blocks inflated on the device (uz_bgzf_inflate, csrc/k_inflate.hip) against zlib on the host: stored, fixed and dynamic DEFLATE
blocks, matches that overlap their own output (runs), codes longer than the direct tables, blocks that end exactly at 64 KiB, empty
blocks (the EOF marker), the generator's BAM (libdeflate's compressor) -- and streams that do not decode are refused, the block named."""
import gzip
import struct
import zlib

import os

import numpy as np
import pytest

from unfazed_amd.engine import UnfazedHipError

pytestmark = pytest.mark.gpu


def bgzf_block(payload: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    body = c.compress(payload) + c.flush()
    bsize = len(body) + 25
    assert bsize <= 65536 + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body
            + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))


def payloads():
    rng = np.random.default_rng(17)
    text = (b"ACGTTGCAAGCT" * 40 + b"\n@read/1 151M * 0 0 " + bytes(rng.integers(33, 74, 151, dtype=np.uint8))) * 60
    out = [b"", b"A", bytes(range(256)) * 3, b"\x00" * 65280, text[:65280], text[:30000],
           bytes(rng.integers(0, 256, 40000, dtype=np.uint8)),          # incompressible: long codes / stored
           bytes(rng.integers(0, 4, 65280, dtype=np.uint8)),            # two-bit entropy: short codes, many matches
           b"ab" * 20000 + b"abc" * 5000 + b"x" * 3000,                 # distances shorter than the match lengths
           bytes(rng.choice(np.arange(256, dtype=np.uint8), 50000, p=np.r_[0.5, np.full(255, 0.5 / 255)]))]  # one cheap symbol, 255 dear ones
    return out


def test_every_block_type_equals_zlib(engine):
    blocks, want = [], []
    for p in payloads():
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if level == 0 and len(p) > 65000:
                continue  # (a stored block of 64 KiB does not fit a BGZF block with its framing)
            blocks.append(bgzf_block(p, level, strategy))
            want.append(p)
    data = b"".join(blocks)
    assert gzip.decompress(data) == b"".join(want)  # (the blocks are what htslib would call a BGZF file)
    got, nb, _ = engine.bgzf_inflate(data)
    assert nb == len(blocks)
    assert bytes(got) == b"".join(want)
    # one block alone, at every alignment of its first byte inside the buffer
    one = bgzf_block(payloads()[4], 6)
    for shift in range(4):
        got, nb, _ = engine.bgzf_inflate(bgzf_block(b"x" * shift, 1) + one) if shift else engine.bgzf_inflate(one)
        assert bytes(got)[-len(payloads()[4]):] == payloads()[4]


def test_the_generators_bam(engine, tmp_path):
    from synth import bigsynth
    from synth.sites_np import make_clusters, make_sites, place_dnms_full
    sc = make_sites(40_000, seed=7, contig_lens=[6e6, 4e6, 2e6])
    dn = place_dnms_full(sc, 40, seed=8, indel_frac=0.2)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=9)
    cfg.n_clusters = cl.n
    for level in (1, 6):
        bam = str(tmp_path / ("kid%d.bam" % level))
        bigsynth.write_bam(bam, cfg, sc, dn, cl, 0, cl.n, level=level)
        data = open(bam, "rb").read()
        got, nb, ms = engine.bgzf_inflate(data, repeat=3)
        want = gzip.decompress(data)
        assert nb > 50 and bytes(got) == want
        print("BAM level %d: %d blocks, %.1f MB -> %.1f MB, %.3f ms per launch = %.1f GB/s of output" % (
            level, nb, len(data) / 1e6, len(want) / 1e6, ms, len(want) / ms / 1e6))


def test_broken_streams_are_refused(engine):
    good = bgzf_block(payloads()[5], 6)
    for breakage in ("flip", "size", "truncated_code"):
        b = bytearray(good)
        if breakage == "flip":
            b[18 + len(b) // 3] ^= 0x55
        elif breakage == "size":
            b[-4:] = struct.pack("<I", len(payloads()[5]) - 7)
        else:
            b[18:22] = b"\xff\xff\xff\xff"  # block type 3 / nonsense lengths
        try:
            got, _, _ = engine.bgzf_inflate(bytes(b))
        except UnfazedHipError as e:
            assert "block 0" in str(e)
            continue
        assert breakage == "flip" and bytes(got) != payloads()[5]  # (a flipped bit can still be a valid stream of the same size: only then)


def test_a_staged_batch_with_its_blocks_inflated_on_the_device(engine, tmp_path):
    """BamSource.select(inflate=engine.inflate_blocks): the blocks a batch's walk reads come back from the device, and the batch is the
    one the host's inflate gives, byte for byte"""
    from synth import bigsynth
    from synth.sites_np import make_clusters, make_sites, place_dnms_full
    from unfazed_amd import io_native
    from unfazed_amd.engine import PinnedPair
    from test_io_stage import assert_same
    sc = make_sites(40_000, seed=7, contig_lens=[6e6, 4e6, 2e6])
    dn = place_dnms_full(sc, 80, seed=8, indel_frac=0.2)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=9)
    cfg.n_clusters = cl.n
    bam = str(tmp_path / "kid.bam")
    bigsynth.write_bam(bam, cfg, sc, dn, cl, 0, cl.n, level=6)
    src = io_native.BamSource(bam, threads=2)
    tid = dn.contig[::2].astype(np.int32)
    lo = (dn.start[::2] - 1).astype(np.int32)
    ex = np.zeros(tid.size, np.uint16)
    plain = src.select(tid, lo, lo + 2, 20, extra=ex)
    pair = PinnedPair()
    for _ in range(2):  # (the second batch re-uses the pinned pair)
        pair.start()
        got = src.select(tid, lo, lo + 2, 20, extra=ex, inflate=engine.inflate_blocks, inflate_alloc=pair.alloc)
        assert_same(got, plain)
        assert got.io_stats["blocks_from_the_device"] == got.io_stats["blocks_inflated"] > 20
    pair.free_all()


@pytest.mark.parametrize("seed", [2024] + [int(x) for x in os.environ.get("UZ_INFLATE_FUZZ_SEEDS", "").split(",") if x])
def test_random_blocks_against_zlib(engine, seed):
    """three hundred blocks of mixed statistics (alphabet size, run lengths, copies of earlier stretches at every distance up to the
    window, sizes from one byte to the 64 KiB limit), every zlib level and strategy in turn (UZ_INFLATE_FUZZ_SEEDS=1,2,...: more seeds)"""
    rng = np.random.default_rng(seed)
    blocks, want = [], []
    strategies = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED)
    for k in range(300):
        n = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 4000, 33000, 65280])) if k % 3 == 0 else int(rng.integers(1, 65281))
        alpha = int(rng.choice([1, 2, 4, 20, 64, 256]))
        p = bytearray(rng.integers(0, alpha, n, dtype=np.uint8).tobytes())
        for _ in range(int(rng.integers(0, 40))):  # copies of earlier stretches: matches at chosen distances, some overlapping their own output
            if n < 8:
                break
            ln = int(rng.integers(3, min(300, n)))
            dst = int(rng.integers(1, n - ln + 1)) if n - ln >= 1 else 0
            dist = int(rng.integers(1, min(dst, 32768) + 1)) if dst >= 1 else 0
            if dist:
                for i in range(ln):
                    p[dst + i] = p[dst + i - dist]
        p = bytes(p)
        level = int(rng.integers(0, 10))
        if level == 0 and n > 65000:
            level = 1
        while True:  # (a payload a fixed or Huffman-only code blows up beyond what a BGZF block can hold is cut down)
            try:
                blocks.append(bgzf_block(p, level, strategies[k % len(strategies)]))
                break
            except AssertionError:
                p = p[: len(p) // 2]
        want.append(p)
    got, nb, _ = engine.bgzf_inflate(b"".join(blocks))
    assert nb == 300
    off = 0
    for k, p in enumerate(want):
        assert bytes(got[off: off + len(p)]) == p, (k, len(p))
        off += len(p)


class _Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, val, nbits):  # LSB first, as DEFLATE packs everything but Huffman codes
        self.v |= (val & ((1 << nbits) - 1)) << self.n
        self.n += nbits

    def code(self, code, nbits):  # a Huffman code: most significant bit first
        for b in range(nbits - 1, -1, -1):
            self.put((code >> b) & 1, 1)

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


def test_a_literal_code_of_one_single_code_is_taken_as_zlib_takes_it(engine):
    """inftrees.c lets an INCOMPLETE literal / length code pass when it is ONE code of one bit ("max != 1" is what it refuses): a dynamic block
    whose only symbol is end-of-block.  No compressor emits it, but the host inflaters decode it -- so must the device (round 3's review)."""
    payload = b"five!"                  # a stored block first, not final ...
    body = b"\x00" + struct.pack("<HH", len(payload), 0xFFFF ^ len(payload)) + payload  # (BFINAL 0, BTYPE 00, padded to the byte)
    w = _Bits()
    w.put(1, 1); w.put(2, 2)            # ... then the final block, dynamic
    w.put(0, 5); w.put(0, 5); w.put(14, 4)  # HLIT 257, HDIST 1, HCLEN 18 code-length codes
    # code-length code, in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1: symbol 18 one bit, symbols 0 and 1 two bits (complete)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1]
    cl = {18: 1, 0: 2, 1: 2}
    for sym in order:
        w.put(cl.get(sym, 0), 3)
    # canonical codes: 18 -> 0, 0 -> 10, 1 -> 11
    w.code(0b0, 1); w.put(127, 7)       # 18: 138 zeros
    w.code(0b0, 1); w.put(107, 7)       # 18: 118 zeros  (256 literals of length 0)
    w.code(0b11, 2)                     # symbol 256 (end of block): length 1
    w.code(0b10, 2)                     # the one distance code: length 0
    w.code(0b0, 1)                      # the data: end of block
    stream = body + w.bytes()
    assert zlib.decompress(stream, -15) == payload  # zlib takes it
    bsize = len(stream) + 25
    block = (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + stream
             + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload)))
    got, nb, _ = engine.bgzf_inflate(block + bgzf_block(b"and a normal block behind it", 6))
    assert nb == 2 and bytes(got) == payload + b"and a normal block behind it"


def test_a_stream_that_never_ends_is_stopped(engine):
    """empty stored blocks without end (BFINAL never set), the last one of block A stepping over A's trailer and B's header into B's
    body, which goes on the same way: the decoder gives up once it has read more than a BGZF block can hold (error 6) -- it does not
    walk on through the buffer and out of it"""
    def frame(body, isize):
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", 0, isize))
    empty = b"\x00\x00\x00\xff\xff"
    a = frame(empty * 13101 + b"\x00" + struct.pack("<HH", 26, 26 ^ 0xFFFF), 100)  # the last stored block: 26 bytes = A's trailer + B's header
    assert len(a) == 65536
    b = frame(empty * 4000, 100)
    good = bgzf_block(b"hello" * 100, 6)
    with pytest.raises(UnfazedHipError, match="block"):
        engine.bgzf_inflate(good + a + b + good * 3)
    out, nb, _ = engine.bgzf_inflate(good * 4)  # and the context is fine afterwards
    assert nb == 4 and out.tobytes() == b"hello" * 400


def test_crc32_kernel_against_zlib(engine):
    """k_bgzf_crc32 (one wavefront per block, 64 slices joined with GF(2) shift matrices) == zlib.crc32 on blocks of every awkward size"""
    import zlib
    rng = np.random.default_rng(77)
    sizes = [0, 1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 1000, 4095, 4096, 4097, 65279, 65280, 65535, 65536] + [int(x) for x in rng.integers(0, 65537, 200)]
    blocks = [rng.integers(0, 256, n, dtype=np.uint8) for n in sizes]
    blocks[7][:] = 0          # all zeros: the shift matrices alone
    blocks[8][:] = 255
    data = np.concatenate(blocks) if blocks else np.zeros(0, np.uint8)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    want = np.array([zlib.crc32(b.tobytes()) & 0xFFFFFFFF for b in blocks], np.uint32)
    assert engine.crc32_blocks(data, off, want) == -1
    for k in (0, 7, 19, len(sizes) - 1):  # one wrong value: that block is named
        w2 = want.copy()
        w2[k] ^= np.uint32(1 << (k % 32))
        assert engine.crc32_blocks(data, off, w2) == k
    d2 = data.copy()  # one flipped bit in the data
    d2[off[19] + 65535] ^= 0x10
    assert engine.crc32_blocks(d2, off, want) == 19
