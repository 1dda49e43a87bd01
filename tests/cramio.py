"""Test-side CRAM 3.0 WRITER (tests only): turns model.Segment records + a reference sequence into a CRAM file and its
.crai, so that unfazed_amd/io_cram.py can be checked by round trips against the BAM decoders on the same records.

It deliberately spreads the data series over every coding the decoder knows: EXTERNAL, HUFFMAN (one symbol with no
bits, and several symbols), BETA / GAMMA / SUBEXP in the core bit stream, BYTE_ARRAY_STOP and BYTE_ARRAY_LEN; and the
external blocks over every block method (raw, gzip, bzip2, lzma, rANS 4x8 order 0 and order 1).  Mates inside a slice
are written as "mate downstream" chains when that reproduces the record's own mate fields, detached otherwise -- the
choice an htslib-style writer makes.  This file and io_cram.py share one author's reading of the format description:
the round trip pins the decoder's internal consistency and its equivalence with the BAM path, not htslib's output.
"""
import bz2
import gzip
import hashlib
import heapq
import lzma
import struct
import zlib

FUNMAP, FMUNMAP, FREVERSE, FMREVERSE, FREAD1 = 4, 8, 16, 32, 64


# ------------------------------------------------------------------ integers
def itf8(v: int) -> bytes:
    v &= 0xFFFFFFFF
    if v < 0x80:
        return bytes([v])
    if v < 0x4000:
        return bytes([0x80 | (v >> 8), v & 0xFF])
    if v < 0x200000:
        return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000:
        return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | (v >> 28), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def ltf8(v: int) -> bytes:
    v &= (1 << 64) - 1
    if v < 0x80:
        return bytes([v])
    for extra in range(1, 8):
        if v < 1 << (7 * (extra + 1)):
            lead = (0xFF << (8 - extra)) & 0xFF
            body = v.to_bytes(extra + 1, "big")
            return bytes([lead | body[0]]) + body[1:]
    return b"\xff" + v.to_bytes(8, "big")


def itf8_array(vals) -> bytes:
    return itf8(len(vals)) + b"".join(itf8(v) for v in vals)


class BitWriter:
    def __init__(self):
        self.acc, self.n = 0, 0

    def put(self, value: int, bits: int):
        if bits:
            self.acc = (self.acc << bits) | (value & ((1 << bits) - 1))
            self.n += bits

    def bytes(self) -> bytes:
        pad = (-self.n) % 8
        return ((self.acc << pad).to_bytes((self.n + pad) // 8, "big")) if self.n else b""


# ------------------------------------------------------------------ rANS 4x8 encoder
def _normalise(counts):
    total = sum(counts)
    f = [0] * 256
    if total == 0:
        return f
    for s, c in enumerate(counts):
        if c:
            f[s] = max(1, c * 4096 // total)
    big = max(range(256), key=lambda s: f[s])
    f[big] += 4096 - sum(f)
    assert f[big] > 0 and sum(f) == 4096
    return f


def _freq_table(f) -> bytes:
    out = bytearray()
    rle = 0
    for j in range(256):
        if not f[j]:
            continue
        if rle:
            rle -= 1
        else:
            out.append(j)
            if j and f[j - 1]:
                r = j + 1
                while r < 256 and f[r]:
                    r += 1
                rle = r - (j + 1)
                out.append(rle)
        if f[j] < 128:
            out.append(f[j])
        else:
            out += bytes([128 | (f[j] >> 8), f[j] & 0xFF])
    out.append(0)
    return bytes(out)


def _cum(f):
    c, x = [0] * 256, 0
    for s in range(256):
        c[s] = x
        x += f[s]
    return c


def rans4x8_encode(data: bytes, order: int) -> bytes:
    n = len(data)
    rev = bytearray()  # bytes in the order they are emitted (the stream is written back to front)
    st = [1 << 23] * 4

    def put(k, f, c):
        x = st[k]
        x_max = ((1 << 23) >> 12 << 8) * f
        while x >= x_max:
            rev.append(x & 0xFF)
            x >>= 8
        st[k] = ((x // f) << 12) + (x % f) + c

    if order == 0:
        counts = [0] * 256
        for b in data:
            counts[b] += 1
        f = _normalise(counts)
        c = _cum(f)
        table = _freq_table(f)
        for i in range(n - 1, -1, -1):
            put(i & 3, f[data[i]], c[data[i]])
    else:
        q = n >> 2
        starts = [0, q, 2 * q, 3 * q]

        def ctx_of(j, k):
            return data[j - 1] if j > starts[k] else 0
        counts = [[0] * 256 for _ in range(256)]
        for k in range(4):
            hi = starts[k] + q if k < 3 else n
            for j in range(starts[k], hi):
                counts[ctx_of(j, k)][data[j]] += 1
        F = {i: _normalise(counts[i]) for i in range(256) if sum(counts[i])}
        C = {i: _cum(F[i]) for i in F}
        tab = bytearray()
        rle = 0
        for i in range(256):
            if i not in F:
                continue
            if rle:
                rle -= 1
            else:
                tab.append(i)
                if i and (i - 1) in F:
                    r = i + 1
                    while r < 256 and r in F:
                        r += 1
                    rle = r - (i + 1)
                    tab.append(rle)
            tab += _freq_table(F[i])
        tab.append(0)
        table = bytes(tab)
        for j in range(n - 1, 4 * q - 1, -1):
            cx = ctx_of(j, 3)
            put(3, F[cx][data[j]], C[cx][data[j]])
        for t in range(q - 1, -1, -1):
            for k in (3, 2, 1, 0):
                j = starts[k] + t
                cx = ctx_of(j, k)
                put(k, F[cx][data[j]], C[cx][data[j]])
    for k in (3, 2, 1, 0):
        rev += struct.pack("<I", st[k])[::-1]
    body = table + bytes(rev[::-1])
    return bytes([order]) + struct.pack("<II", len(body), n) + body


# ------------------------------------------------------------------ blocks / containers
RAW, GZIP, BZIP2, LZMA, RANS0, RANS1 = 0, 1, 2, 3, 40, 41


def block(method: int, ctype: int, cid: int, data: bytes) -> bytes:
    if not data:
        method = RAW
    if method == RAW:
        comp, m = data, 0
    elif method == GZIP:
        comp, m = gzip.compress(data, 6), 1
    elif method == BZIP2:
        comp, m = bz2.compress(data), 2
    elif method == LZMA:
        comp, m = lzma.compress(data), 3
    else:
        comp, m = rans4x8_encode(data, method - 40), 4
    body = bytes([m, ctype]) + itf8(cid) + itf8(len(comp)) + itf8(len(data)) + comp
    return body + struct.pack("<I", zlib.crc32(body) & 0xFFFFFFFF)


def container(ref, start, span, n_records, counter, bases, n_blocks, landmarks, body: bytes) -> bytes:
    head = struct.pack("<i", len(body)) + itf8(ref) + itf8(start) + itf8(span) + itf8(n_records) + ltf8(counter) + ltf8(bases) + \
        itf8(n_blocks) + itf8_array(landmarks)
    return head + struct.pack("<I", zlib.crc32(head) & 0xFFFFFFFF) + body


# ------------------------------------------------------------------ encodings
def enc_external(cid):
    p = itf8(cid)
    return itf8(1) + itf8(len(p)) + p


def enc_huffman(syms, lens):
    p = itf8_array(syms) + itf8_array(lens)
    return itf8(3) + itf8(len(p)) + p


def enc_beta(offset, bits):
    p = itf8(offset) + itf8(bits)
    return itf8(6) + itf8(len(p)) + p


def enc_gamma(offset):
    p = itf8(offset)
    return itf8(9) + itf8(len(p)) + p


def enc_subexp(offset, k):
    p = itf8(offset) + itf8(k)
    return itf8(7) + itf8(len(p)) + p


def enc_stop(stop, cid):
    p = bytes([stop]) + itf8(cid)
    return itf8(5) + itf8(len(p)) + p


def enc_len(len_enc, val_enc):
    p = len_enc + val_enc
    return itf8(4) + itf8(len(p)) + p


def huffman_lengths(counts: dict) -> dict:
    if len(counts) == 1:
        return {next(iter(counts)): 0}
    heap = [(c, i, [s]) for i, (s, c) in enumerate(sorted(counts.items()))]
    heapq.heapify(heap)
    lens = {s: 0 for s in counts}
    k = len(heap)
    while len(heap) > 1:
        a = heapq.heappop(heap)
        b = heapq.heappop(heap)
        for s in a[2] + b[2]:
            lens[s] += 1
        heapq.heappush(heap, (a[0] + b[0], k, a[2] + b[2]))
        k += 1
    return lens


def canonical_codes(lens: dict) -> dict:
    order = sorted((ln, s) for s, ln in lens.items())
    codes, code, prev = {}, 0, order[0][0]
    for ln, s in order:
        code <<= ln - prev
        codes[s] = (code, ln)
        code += 1
        prev = ln
    return codes


# ------------------------------------------------------------------ records -> slices
_REF_OPS, _QUERY_OPS = (0, 2, 3, 7, 8), (0, 1, 4, 7, 8)
SM_DEFAULT = bytes([0x1B] * 5)  # for every reference base the other four bases in A C G T N order get codes 0 1 2 3
SM_OTHER = bytes([0xE4, 0x93, 0x4E, 0x39, 0x1B])


def _subst_code(sm: bytes, ref: int, alt: int) -> int:
    r = b"ACGTN".index(ref)
    others = [x for x in b"ACGTN" if x != ref]
    return (sm[r] >> (6 - 2 * others.index(alt))) & 3


class _Streams:
    def __init__(self):
        self.ext = {}
        self.core = BitWriter()

    def e(self, cid) -> bytearray:
        return self.ext.setdefault(cid, bytearray())


def _aend1(s) -> int:
    """1-based inclusive alignment end as the decoder defines it"""
    if s.flag & FUNMAP:
        return s.pos + 1
    rl = sum(ln for op, ln in s.cigar if op in _REF_OPS)
    return s.pos + max(rl, 1) if rl else s.pos + 1


def _derive_chain(chain):
    """mate fields a decoder gives the records of a mate-downstream chain -> [(flag bits to add, mtid, mpos, tlen)]"""
    left = min(s.pos + 1 for s in chain)
    right = max(_aend1(s) for s in chain)
    left_cnt = sum(1 for s in chain if s.pos + 1 == left)
    same = all(s.tid == chain[0].tid for s in chain)
    out = []
    for i, s in enumerate(chain):
        m = chain[(i + 1) % len(chain)]
        t = right - left + 1
        tlen = (t if (s.pos + 1 == left and (left_cnt == 1 or s.flag & FREAD1)) else -t) if same else 0
        add = 0
        if m.flag & FUNMAP:
            add |= FMUNMAP
            tlen = 0
        if s.flag & FUNMAP:
            tlen = 0
        if m.flag & FREVERSE:
            add |= FMREVERSE
        out.append((add, m.tid, m.pos, tlen))
    return out


def write_cram(path, contigs, segs, ref, records_per_slice=700, slices_per_container=2, keep_names=True, sm=SM_DEFAULT,
               embed_ref=False, multi_ref_slices=False, base_stretches=False, methods=(RAW, GZIP, BZIP2, LZMA, RANS0, RANS1),
               md5=True, write_index=True):
    """contigs: [(name, length)]; segs: model.Segment in file order; ref: {contig name: bytes (upper case)}."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(
        "@SQ\tSN:%s\tLN:%d\tM5:%s\n" % (n, ln, hashlib.md5(ref[n]).hexdigest() if n in ref else "0" * 32) for n, ln in contigs)
    hdr_block = block(GZIP, 0, 0, struct.pack("<i", len(text)) + text.encode())
    out = bytearray(b"CRAM\x03\x00" + b"unfazed-amd-test\0\0\0\0")
    out += container(0, 0, 0, 0, 0, 0, 1, [0], hdr_block)
    names = [n for n, _ in contigs]
    crai = []
    # cut into slices: a slice holds records of one contig unless multi_ref_slices
    slices, cur = [], []
    for s in segs:
        if cur and (len(cur) >= records_per_slice or (not multi_ref_slices and s.tid != cur[0].tid)):
            slices.append(cur)
            cur = []
        cur.append(s)
    if cur:
        slices.append(cur)
    counter = 0
    for c0 in range(0, len(slices), slices_per_container):
        group = slices[c0: c0 + slices_per_container]
        rl_counts = {}
        for sl in group:
            for s in sl:
                rl = len(s.seq) if s.seq else sum(ln for op, ln in s.cigar if op in _QUERY_OPS)
                rl_counts[rl] = rl_counts.get(rl, 0) + 1
        rl_huff = len(rl_counts) <= 8
        rl_codes = canonical_codes(huffman_lengths(rl_counts)) if rl_huff else None
        series = {
            "BF": enc_external(1), "CF": enc_external(2), "RI": enc_external(3),
            "RL": enc_huffman([s for s in sorted(rl_codes)], [rl_codes[s][1] for s in sorted(rl_codes)]) if rl_huff else enc_external(4),
            "AP": enc_external(5), "RG": enc_huffman([-1], [0]), "RN": enc_stop(0, 6), "MF": enc_external(7), "NS": enc_external(8),
            "NP": enc_external(9), "TS": enc_external(10), "NF": enc_external(11), "TL": enc_beta(0, 1), "FN": enc_external(12),
            "FC": enc_external(13), "FP": enc_gamma(1), "BS": enc_external(15), "IN": enc_stop(0, 16),
            "SC": enc_len(enc_external(17), enc_external(18)), "BA": enc_external(19), "QS": enc_external(20), "DL": enc_subexp(0, 2),
            "RS": enc_external(21), "HC": enc_external(22), "PD": enc_external(23), "MQ": enc_beta(0, 8),
            "BB": enc_len(enc_external(24), enc_external(25)), "QQ": enc_len(enc_gamma(0), enc_external(27)),
        }
        sa_key = (ord("S") << 16) | (ord("A") << 8) | ord("Z")
        pres = [b"RN" + bytes([1 if keep_names else 0]), b"AP\x01", b"RR" + bytes([0 if embed_ref else 1]), b"SM" + sm,
                b"TD" + itf8(len(b"\0SAZ\0")) + b"\0SAZ\0"]
        pm = itf8(len(pres)) + b"".join(pres)
        dm = itf8(len(series)) + b"".join(k.encode() + v for k, v in series.items())
        tm = itf8(1) + itf8(sa_key) + enc_stop(9, sa_key)
        ch = block(RAW, 1, 0, itf8(len(pm)) + pm + itf8(len(dm)) + dm + itf8(len(tm)) + tm)
        body = bytearray(ch)
        landmarks, n_blocks, n_rec_c, bases_c = [], 1, 0, 0
        c_refs = set()
        slice_meta = []
        for sl in group:
            st = _Streams()
            tids = {s.tid for s in sl}
            sref = sl[0].tid if len(tids) == 1 else -2
            c_refs.add(sref)
            if sref >= 0:
                start1 = min(s.pos + 1 for s in sl)
                end1 = max(_aend1(s) for s in sl)
                span = end1 - start1 + 1
            else:
                start1, span = 0, 0
            # mate chains inside the slice
            by_name = {}
            for i, s in enumerate(sl):
                by_name.setdefault(s.qname, []).append(i)
            link = {}  # record -> next record of its chain
            attached = set()
            for nm, idx in by_name.items():
                if len(idx) < 2:
                    continue
                chain = [sl[i] for i in idx]
                ok = all((s.flag & (FMUNMAP | FMREVERSE)) == add and s.mtid == mt and s.mpos == mp and s.tlen == tl
                         for s, (add, mt, mp, tl) in zip(chain, _derive_chain(chain)))
                if ok:
                    for a, b in zip(idx[:-1], idx[1:]):
                        link[a] = b
                    attached.update(idx)
            last = start1
            for i, s in enumerate(sl):
                mapped = not s.flag & FUNMAP
                no_seq = not s.seq
                rl = len(s.seq) if s.seq else sum(ln for op, ln in s.cigar if op in _QUERY_OPS)
                cfl = (1 if s.qual is not None else 0) | (8 if no_seq else 0)
                bf = s.flag
                detached = False
                if i in link:
                    cfl |= 4
                    bf &= ~(FMUNMAP | FMREVERSE)
                elif i in attached:
                    bf &= ~(FMUNMAP | FMREVERSE)  # the last record of a chain: everything comes from the chain
                elif s.flag & 1 or s.mtid != -1 or s.mpos != -1 or s.tlen != 0 or s.flag & (FMUNMAP | FMREVERSE):
                    cfl |= 2
                    detached = True
                    bf &= ~(FMUNMAP | FMREVERSE)
                st.e(1).extend(itf8(bf))
                st.e(2).extend(itf8(cfl))
                if sref == -2:
                    st.e(3).extend(itf8(s.tid))
                if rl_huff:
                    st.core.put(*rl_codes[rl])
                else:
                    st.e(4).extend(itf8(rl))
                st.e(5).extend(itf8(s.pos + 1 - last))
                last = s.pos + 1
                if keep_names:
                    st.e(6).extend(s.qname.encode() + b"\0")
                if detached:
                    st.e(7).extend(itf8((1 if s.flag & FMREVERSE else 0) | (2 if s.flag & FMUNMAP else 0)))
                    if not keep_names:
                        st.e(6).extend(s.qname.encode() + b"\0")
                    st.e(8).extend(itf8(s.mtid))
                    st.e(9).extend(itf8(s.mpos + 1))
                    st.e(10).extend(itf8(s.tlen))
                elif i in link:
                    st.e(11).extend(itf8(link[i] - i - 1))
                st.core.put(1 if s.has_sa else 0, 1)
                if s.has_sa:
                    st.e(sa_key).extend(b"x,1,+,10M,60,0;\0\t")
                if mapped:
                    feats = []  # (read position 1-based, code, payload writer)
                    rpos, spos = s.pos, 0
                    rseq = ref.get(names[s.tid], b"") if 0 <= s.tid < len(names) else b""
                    sq = s.seq.encode() if s.seq else b"N" * rl
                    for op, ln in s.cigar:
                        if op in (0, 7, 8):
                            k = 0
                            while k < ln:
                                rb = rseq[rpos + k] if rpos + k < len(rseq) else 78
                                b = sq[spos + k]
                                if no_seq or b == rb:
                                    k += 1
                                    continue
                                run = 1
                                while base_stretches and k + run < ln and sq[spos + k + run] != (rseq[rpos + k + run] if rpos + k + run < len(rseq) else 78):
                                    run += 1
                                if run >= 3:
                                    feats.append((spos + k + 1, b"b", sq[spos + k: spos + k + run]))
                                    k += run
                                    continue
                                if rb in b"ACGTN" and b in b"ACGTN":
                                    feats.append((spos + k + 1, b"X", bytes([_subst_code(sm, rb, b)])))
                                else:
                                    feats.append((spos + k + 1, b"B", bytes([b, s.qual[spos + k] if s.qual is not None else 0xFF])))
                                k += 1
                            rpos += ln
                            spos += ln
                        elif op == 1:
                            if ln == 1 and not no_seq:
                                feats.append((spos + 1, b"i", sq[spos: spos + 1]))
                            else:
                                feats.append((spos + 1, b"I", sq[spos: spos + ln]))
                            spos += ln
                        elif op == 4:
                            feats.append((spos + 1, b"S", sq[spos: spos + ln]))
                            spos += ln
                        elif op == 2:
                            feats.append((spos + 1, b"D", ln))
                            rpos += ln
                        elif op == 3:
                            feats.append((spos + 1, b"N", ln))
                            rpos += ln
                        elif op == 5:
                            feats.append((spos + 1, b"H", ln))
                        elif op == 6:
                            feats.append((spos + 1, b"P", ln))
                    st.e(12).extend(itf8(len(feats)))
                    prev = 0
                    for fp, code, val in feats:
                        st.e(13).extend(code)
                        d = fp - prev
                        prev = fp
                        nb = (d + 1).bit_length() - 1  # GAMMA, offset 1: nb zeros, then value + 1 in nb + 1 bits
                        st.core.put(0, nb)
                        st.core.put(d + 1, nb + 1)
                        if code == b"X":
                            st.e(15).extend(val)
                        elif code == b"B":
                            st.e(19).append(val[0])
                            st.e(20).append(val[1])
                        elif code == b"b":
                            st.e(24).extend(itf8(len(val)))
                            st.e(25).extend(val)
                        elif code == b"I":
                            st.e(16).extend(val + b"\0")
                        elif code == b"i":
                            st.e(19).extend(val)
                        elif code == b"S":
                            st.e(17).extend(itf8(len(val)))
                            st.e(18).extend(val)
                        elif code == b"D":
                            v = val  # SUBEXP, offset 0, k = 2
                            if v < 4:
                                st.core.put(0, 1)
                                st.core.put(v, 2)
                            else:
                                b_ = v.bit_length() - 1
                                u = b_ - 2 + 1
                                st.core.put((1 << u) - 1, u)
                                st.core.put(0, 1)
                                st.core.put(v & ((1 << b_) - 1), b_)
                        elif code == b"N":
                            st.e(21).extend(itf8(val))
                        elif code == b"H":
                            st.e(22).extend(itf8(val))
                        elif code == b"P":
                            st.e(23).extend(itf8(val))
                    st.core.put(s.mapq, 8)
                    if s.qual is not None:
                        st.e(20).extend(bytes(s.qual))
                else:
                    if not no_seq:
                        st.e(19).extend(s.seq.encode())
                    if s.qual is not None:
                        st.e(20).extend(bytes(s.qual))
                bases_c += rl
            ids = sorted(st.ext)
            embedded = -1
            blocks = []
            if embed_ref and sref >= 0:
                embedded = 30
                st.ext[30] = bytearray(ref[names[sref]][start1 - 1: start1 - 1 + span])
                ids = sorted(st.ext)
            digest = b"\0" * 16
            if md5 and sref >= 0 and not embed_ref:
                digest = hashlib.md5(ref[names[sref]][start1 - 1: start1 - 1 + span]).digest()
            blocks.append(block(RAW, 5, 0, st.core.bytes()))
            for cid in ids:
                blocks.append(block(methods[cid % len(methods)], 4, cid, bytes(st.ext[cid])))
            sh = itf8(sref) + itf8(start1) + itf8(span) + itf8(len(sl)) + ltf8(counter) + itf8(len(blocks)) + itf8_array(ids) + \
                itf8(embedded) + digest
            lm = len(body)
            landmarks.append(lm)
            body += block(RAW, 2, 0, sh)
            for b in blocks:
                body += b
            n_blocks += 1 + len(blocks)
            per_ref = {}
            for s in sl:
                a, b = s.pos + 1, _aend1(s)
                lo, hi = per_ref.get(s.tid, (a, b))
                per_ref[s.tid] = (min(lo, a), max(hi, b))
            slice_meta.append((lm, len(body) - lm, per_ref))
            counter += len(sl)
            n_rec_c += len(sl)
        cref = next(iter(c_refs)) if len(c_refs) == 1 else -2
        if cref >= 0:
            cs = min(s.pos + 1 for sl in group for s in sl)
            ce = max(_aend1(s) for sl in group for s in sl)
            cstart, cspan = cs, ce - cs + 1
        else:
            cstart, cspan = 0, 0
        coff = len(out)
        out += container(cref, cstart, cspan, n_rec_c, counter - n_rec_c, bases_c, n_blocks, landmarks, bytes(body))
        for lm, size, per_ref in slice_meta:
            for t in sorted(per_ref):
                a, b = per_ref[t]
                crai.append("%d\t%d\t%d\t%d\t%d\t%d\n" % (t, a, b - a + 1, coff, lm, size))
    eof_block = block(RAW, 1, 0, b"\x01\x00\x01\x00\x01\x00")
    out += container(-1, 4542278, 0, 0, 0, 0, 1, [], eof_block)
    with open(path, "wb") as fh:
        fh.write(bytes(out))
    if write_index:
        with gzip.open(path + ".crai", "wt") as fh:
            fh.write("".join(crai))


def write_fasta(path, ref: dict, order, width=60, index=True):
    fai = []
    with open(path, "wb") as fh:
        for name in order:
            seq = ref[name]
            fh.write(b">" + name.encode() + b" test contig\n")
            off = fh.tell()
            for i in range(0, len(seq), width):
                fh.write(seq[i: i + width] + b"\n")
            fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), off, width, width + 1))
    if index:
        with open(path + ".fai", "w") as fh:
            fh.write("".join(fai))
