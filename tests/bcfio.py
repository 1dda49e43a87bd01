"""Minimal BCF2 writer for tests (no bcftools / htslib in the image): the same records tests/filesio.vcf_text
writes as text, in the binary form, BGZF framed.  FORMAT fields GT:AD:GQ, INFO END (Integer) / SVTYPE (String).
Test infrastructure."""
import struct

from unfazed_amd.io_bam import _bgzf_block

INT8_MISSING, INT8_EOV = 0x80, 0x81
INT16_MISSING = 0x8000
INT32_MISSING = 0x80000000
FLOAT_MISSING = 0x7F800001


def _typed_len(n, t):
    if n < 15:
        return bytes([(n << 4) | t])
    return bytes([0xF0 | t]) + _typed_int(n)


def _typed_int(v):
    if -120 <= v <= 127:
        return bytes([0x11]) + struct.pack("<b", v)
    if -32000 <= v <= 32767:
        return bytes([0x12]) + struct.pack("<h", v)
    return bytes([0x13]) + struct.pack("<i", v)


def _typed_str(s):
    b = s.encode()
    return _typed_len(len(b), 7) + b


def bcf_bytes(samples, records, contigs, use_idx=False, int16_depths=False):
    dict_ids = ["PASS", "SVTYPE", "END", "GT", "AD", "GQ"]
    idx = (lambda k: ",IDX=%d" % k) if use_idx else (lambda k: "")
    lines = ["##fileformat=VCFv4.2", '##FILTER=<ID=PASS,Description="All filters passed"%s>' % idx(0)]
    lines += ["##contig=<ID=%s%s>" % (c, idx(i)) for i, c in enumerate(contigs)]
    lines += ['##INFO=<ID=SVTYPE,Number=1,Type=String,Description="sv type"%s>' % idx(1),
              '##INFO=<ID=END,Number=1,Type=Integer,Description="end"%s>' % idx(2),
              '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype"%s>' % idx(3),
              '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths"%s>' % idx(4),
              '##FORMAT=<ID=GQ,Number=1,Type=Float,Description="Genotype quality"%s>' % idx(5)]
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples))
    text = ("\n".join(lines) + "\n").encode() + b"\0"
    out = bytearray(b"BCF\x02\x02" + struct.pack("<I", len(text)) + text)
    ns = len(samples)
    for r in records:
        alleles = [r.ref] + list(r.alts)
        info = b""
        n_info = 0
        for k, v in r.info.items():
            if k == "END":
                info += _typed_int(dict_ids.index("END")) + _typed_int(int(v))
                n_info += 1
            elif k == "SVTYPE":
                info += _typed_int(dict_ids.index("SVTYPE")) + _typed_str(str(v))
                n_info += 1
        rlen = (r.end - r.start) if r.end is not None else len(r.ref)
        shared = struct.pack("<iiifII", contigs.index(r.chrom), r.start, rlen, 50.0, (len(alleles) << 16) | n_info, (3 << 24) | ns)
        shared += bytes([0x07])  # ID: empty string -> '.'
        for a in alleles:
            shared += _typed_str(a)
        shared += bytes([0x11, 0x00])  # FILTER: PASS
        shared += info
        # GT (2 x int8 per sample)
        gt = bytearray()
        for g in r.gt_types:
            a, b = {0: (0, 0), 1: (0, 1), 2: (-1, -1), 3: (1, 1)}[int(g)]
            gt += bytes([((a + 1) << 1) & 0xFF, ((b + 1) << 1) & 0xFF])
        indiv = _typed_int(dict_ids.index("GT")) + bytes([(2 << 4) | 1]) + bytes(gt)
        # AD (R values per sample)
        nad = len(alleles)
        ad = bytearray()
        for i in range(ns):
            rd, al = int(r.ref_depths[i]), int(r.alt_depths[i])
            vals = [rd, al] + [0] * (nad - 2)
            if rd < 0 and al < 0:  # "." in the text form: missing, then end-of-vector
                if int16_depths:
                    ad += struct.pack("<H", INT16_MISSING) + struct.pack("<H", INT16_MISSING + 1) * (nad - 1)
                else:
                    ad += struct.pack("<I", INT32_MISSING) + struct.pack("<I", INT32_MISSING + 1) * (nad - 1)
                continue
            for v in vals[:nad]:
                if int16_depths:
                    ad += struct.pack("<H", INT16_MISSING) if v < 0 else struct.pack("<h", v)
                else:
                    ad += struct.pack("<I", INT32_MISSING) if v < 0 else struct.pack("<i", v)
        indiv += _typed_int(dict_ids.index("AD")) + _typed_len(nad, 2 if int16_depths else 3) + bytes(ad)
        gq = bytearray()
        for i in range(ns):
            q = float(r.gt_quals[i])
            gq += struct.pack("<I", FLOAT_MISSING) if q < 0 else struct.pack("<f", q)
        indiv += _typed_int(dict_ids.index("GQ")) + bytes([(1 << 4) | 5]) + bytes(gq)
        out += struct.pack("<II", len(shared), len(indiv)) + shared + indiv
    return bytes(out)


def write_bcf(path, samples, records, contigs, **kw):
    data = bcf_bytes(samples, records, contigs, **kw)
    with open(path, "wb") as fh:
        for i in range(0, len(data), 60000):
            fh.write(_bgzf_block(data[i: i + 60000]))
        fh.write(_bgzf_block(b""))
