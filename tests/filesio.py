"""Writes a synth.small dataset as real files (sites VCF, DNM VCF / BED, PED, BAM) so the
file decoders and the command line can be exercised end to end.  Test infrastructure."""
import gzip
import os

from unfazed_amd.io_bam import write_bam

GT_TEXT = {0: "0/0", 1: "0/1", 2: "./.", 3: "1/1"}


def _num(x):
    if x is None or x < 0:
        return "."
    return ("%g" % x)


def vcf_text(samples, records, contigs):
    lines = ["##fileformat=VCFv4.2"]
    lines += ["##contig=<ID=%s>" % c for c in contigs]
    lines += ['##INFO=<ID=SVTYPE,Number=1,Type=String,Description="sv type">',
              '##INFO=<ID=END,Number=1,Type=Integer,Description="end">',
              '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
              '##FORMAT=<ID=AD,Number=R,Type=Integer,Description="Allelic depths">',
              '##FORMAT=<ID=GQ,Number=1,Type=Float,Description="Genotype quality">']
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples))
    for r in records:
        info = ";".join("%s=%s" % (k, v) for k, v in r.info.items()) or "."
        cols = []
        for i in range(len(samples)):
            rd, ad = r.ref_depths[i], r.alt_depths[i]
            adtxt = "." if (rd < 0 and ad < 0) else ",".join([_num(rd), _num(ad)] + ["0"] * (len(r.alts) - 1))
            cols.append("%s:%s:%s" % (GT_TEXT[int(r.gt_types[i])], adtxt, _num(float(r.gt_quals[i]))))
        lines.append("\t".join([r.chrom, str(r.start + 1), ".", r.ref, ",".join(r.alts) if r.alts else ".", "50", "PASS",
                                info, "GT:AD:GQ"] + cols))
    return "\n".join(lines) + "\n"


def dump_dataset(ds, outdir, contig_len=10_000_000):
    os.makedirs(outdir, exist_ok=True)
    paths = {}
    paths["sites"] = os.path.join(outdir, "sites.vcf.gz")
    with gzip.open(paths["sites"], "wt") as fh:
        fh.write(vcf_text(ds.samples, ds.sites, ds.contigs))
    # DNM VCF: the DNMs' own records (first record at each DNM position)
    seen = set()
    dn_recs = []
    want = {(d["chrom"], d["start"]) for d in ds.dnms}
    for r in ds.sites:
        k = (r.chrom, r.start)
        if k in want and k not in seen:
            seen.add(k)
            dn_recs.append(r)
    paths["dnm_vcf"] = os.path.join(outdir, "dnms.vcf")
    with open(paths["dnm_vcf"], "w") as fh:
        fh.write(vcf_text(ds.samples, dn_recs, ds.contigs))
    paths["dnm_bed"] = os.path.join(outdir, "dnms.bed")
    with open(paths["dnm_bed"], "w") as fh:
        fh.write("#chrom\tstart\tend\tkid_id\tvar_type\n")
        for d in ds.dnms:
            fh.write("%s\t%d\t%d\t%s\t%s\n" % (d["chrom"], d["start"], d["end"], d["kid"], "SNV"))
    paths["ped"] = os.path.join(outdir, "trio.ped")
    with open(paths["ped"], "w") as fh:
        fh.write("#Family-ID\tIndividual-ID\tPaternal-ID\tMaternal-ID\tGender\n")
        for kid, p in ds.pedigrees.items():
            fh.write("F\t%s\t%s\t%s\t%s\n" % (kid, p["dad"], p["mom"], p["sex"]))
            fh.write("F\t%s\t0\t0\t1\nF\t%s\t0\t0\t2\n" % (p["dad"], p["mom"]))
    paths["bams"] = {}
    for kid, segs in ds.reads.items():
        paths["bams"][kid] = os.path.join(outdir, "%s.bam" % kid)
        write_bam(paths["bams"][kid], [(c, contig_len) for c in ds.contigs], segs)
    return paths


# ---------------------------------------------------------------------------- BAI writer (tests)
def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def write_bai(bam_path, bai_path=None):
    """A BAI for a coordinate-sorted BAM, built the way samtools index does (SAM spec 5.2): bins with merged chunks and
    the 16 kb linear index (empty windows take the following window's offset).  Reads the BAM back with the Python reader."""
    import struct
    import zlib
    raw = open(bam_path, "rb").read()
    # BGZF blocks: compressed offset and inflated size of each
    blocks, p, u = [], 0, 0
    data = bytearray()
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1  # the writer puts BC first
        payload = zlib.decompress(raw[p + 12 + xlen: p + bsize - 8], -15)
        blocks.append((p, u, len(payload)))
        data += payload
        p += bsize
        u += len(payload)

    def voff(uo):  # virtual offset of the byte at inflated offset uo
        import bisect
        k = bisect.bisect_right([b[1] for b in blocks], uo) - 1
        while k + 1 < len(blocks) and blocks[k][2] == 0:
            k += 1
        if uo - blocks[k][1] >= blocks[k][2] and k + 1 < len(blocks):  # the end of a block is the start of the next
            k += 1
        return (blocks[k][0] << 16) | (uo - blocks[k][1])

    l_text = struct.unpack_from("<i", data, 4)[0]
    off = 8 + l_text
    n_ref = struct.unpack_from("<i", data, off)[0]
    off += 4
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", data, off)[0]
        off += 4 + ln + 4
    bins = [dict() for _ in range(n_ref)]
    linear = [dict() for _ in range(n_ref)]
    while off + 4 <= len(data):
        bs = struct.unpack_from("<i", data, off)[0]
        tid, pos, l_name, mapq, bn, ncig, flag, l_seq = struct.unpack_from("<iiBBHHHi", data, off + 4)
        q = off + 4 + 32 + l_name
        rl = 0
        for k in range(ncig):
            v = struct.unpack_from("<I", data, q + 4 * k)[0]
            if (v & 15) in (0, 2, 3, 7, 8):
                rl += v >> 4
        end = pos + 1 if (flag & 4) or ncig == 0 else pos + max(rl, 1)
        v0, v1 = voff(off), voff(off + 4 + bs)
        if tid >= 0:
            b = _reg2bin(pos, end)
            ch = bins[tid].setdefault(b, [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            for w in range(pos >> 14, ((end - 1) >> 14) + 1):
                if w not in linear[tid]:
                    linear[tid][w] = v0
        off += 4 + bs
    out = bytearray(b"BAI\x01" + struct.pack("<i", n_ref))
    for t in range(n_ref):
        out += struct.pack("<i", len(bins[t]))
        for b in sorted(bins[t]):
            out += struct.pack("<Ii", b, len(bins[t][b]))
            for v0, v1 in bins[t][b]:
                out += struct.pack("<QQ", v0, v1)
        n_intv = (max(linear[t]) + 1) if linear[t] else 0
        lin = [linear[t].get(w, 0) for w in range(n_intv)]
        for w in range(n_intv - 2, -1, -1):
            if lin[w] == 0:
                lin[w] = lin[w + 1]
        out += struct.pack("<i", n_intv)
        for v in lin:
            out += struct.pack("<Q", v)
    bai_path = bai_path or bam_path + ".bai"
    with open(bai_path, "wb") as fh:
        fh.write(bytes(out))
    return bai_path


# ---------------------------------------------------------------------------- BGZF text + tabix index writer (tests)
def write_bgzf_text(path, text: str, block_bytes=6000):
    """text -> BGZF file (small blocks, so that a region touches few of many)"""
    from unfazed_amd.io_bam import _bgzf_block
    data = text.encode()
    with open(path, "wb") as fh:
        for i in range(0, len(data), block_bytes):
            fh.write(_bgzf_block(data[i: i + block_bytes]))
        fh.write(_bgzf_block(b""))


def write_tbi(vcf_gz_path, tbi_path=None):
    """A tabix index for a BGZF-compressed, position-sorted VCF, built the way `tabix -p vcf` does (the binning scheme and
    linear index of the SAM spec; the end of a record is POS + len(REF) - 1, or INFO/END when it reaches further)."""
    import struct
    import zlib
    from unfazed_amd.io_bam import _bgzf_block
    raw = open(vcf_gz_path, "rb").read()
    blocks, p, u = [], 0, 0
    data = bytearray()
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        payload = zlib.decompress(raw[p + 12 + xlen: p + bsize - 8], -15)
        blocks.append((p, u, len(payload)))
        data += payload
        p += bsize
        u += len(payload)
    starts = [b[1] for b in blocks]

    def voff(uo):
        import bisect
        k = bisect.bisect_right(starts, uo) - 1
        while k + 1 < len(blocks) and uo - blocks[k][1] >= blocks[k][2]:
            k += 1
        return (blocks[k][0] << 16) | (uo - blocks[k][1])

    names, bins, linear = [], [], []
    off = 0
    n = len(data)
    while off < n:
        e = data.find(b"\n", off)
        if e < 0:
            e = n
        line = bytes(data[off:e])
        if line and not line.startswith(b"#"):
            f = line.split(b"\t")
            chrom, pos0 = f[0].decode(), int(f[1]) - 1
            end = pos0 + max(1, len(f[3]))
            if len(f) > 7:
                for kv in f[7].split(b";"):
                    if kv.startswith(b"END="):
                        try:
                            end = max(end, int(kv[4:]))
                        except ValueError:
                            pass
            if not names or names[-1] != chrom:
                assert chrom not in names, "records are not grouped by contig"
                names.append(chrom)
                bins.append({})
                linear.append({})
            v0, v1 = voff(off), voff(min(e + 1, n))
            ch = bins[-1].setdefault(_reg2bin(pos0, end), [])
            if ch and ch[-1][1] == v0:
                ch[-1][1] = v1
            else:
                ch.append([v0, v1])
            for w in range(pos0 >> 14, ((end - 1) >> 14) + 1):
                linear[-1].setdefault(w, v0)
        off = e + 1
    nm = b"".join(x.encode() + b"\0" for x in names)
    out = bytearray(b"TBI\x01" + struct.pack("<iiiiiiii", len(names), 2, 1, 2, 0, ord("#"), 0, len(nm)) + nm)
    for t in range(len(names)):
        out += struct.pack("<i", len(bins[t]))
        for b in sorted(bins[t]):
            out += struct.pack("<Ii", b, len(bins[t][b]))
            for v0, v1 in bins[t][b]:
                out += struct.pack("<QQ", v0, v1)
        n_intv = (max(linear[t]) + 1) if linear[t] else 0
        lin = [linear[t].get(w, 0) for w in range(n_intv)]
        for w in range(n_intv - 2, -1, -1):
            if lin[w] == 0:
                lin[w] = lin[w + 1]
        out += struct.pack("<i", n_intv)
        for v in lin:
            out += struct.pack("<Q", v)
    tbi_path = tbi_path or vcf_gz_path + ".tbi"
    with open(tbi_path, "wb") as fh:
        blob = bytes(out)
        for i in range(0, len(blob), 60000):
            fh.write(_bgzf_block(blob[i: i + 60000]))
        fh.write(_bgzf_block(b""))
    return tbi_path


def _inflated_blocks(path):
    """(compressed offset, offset in the inflated stream, inflated length) of every BGZF block + the inflated stream"""
    import struct
    import zlib
    raw = open(path, "rb").read()
    blocks, p, u = [], 0, 0
    data = bytearray()
    while p < len(raw):
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        payload = zlib.decompress(raw[p + 12 + xlen: p + bsize - 8], -15)
        blocks.append((p, u, len(payload)))
        data += payload
        p += bsize
        u += len(payload)
    return blocks, bytes(data)


def write_csi(path, csi_path=None, min_shift=14, depth=5):
    """A CSI index (SAM/CSI spec, CSIv1) for a BGZF-compressed, position-sorted VCF -- as `tabix -C` -- or for a BCF -- as
    `bcftools index`: bins of a scheme min_shift / depth wide, a left-most offset per bin (the offset of the first record that overlaps
    the bin's first window or starts behind it: what htslib derives from its linear index), the pseudo-bin with the counts.  The
    references of a BCF are its header's contig ids; a text file's names travel in the aux block (the tabix header).  Test infrastructure."""
    import bisect
    import struct
    from unfazed_amd.io_bam import _bgzf_block
    blocks, data = _inflated_blocks(path)
    starts = [b[1] for b in blocks]

    def voff(uo):
        k = bisect.bisect_right(starts, uo) - 1
        while k + 1 < len(blocks) and uo - blocks[k][1] >= blocks[k][2]:
            k += 1
        return (blocks[k][0] << 16) | (uo - blocks[k][1])

    recs = []  # (reference, pos0, end, v0, v1) in file order
    names = []
    is_bcf = data[:5] == b"BCF\x02\x02"
    if is_bcf:
        off = 9 + struct.unpack_from("<I", data, 5)[0]
        while off + 8 <= len(data):
            ls, li = struct.unpack_from("<II", data, off)
            chrom, pos0, rlen = struct.unpack_from("<iii", data, off + 8)
            recs.append((chrom, pos0, pos0 + max(1, rlen), voff(off), voff(off + 8 + ls + li) if off + 8 + ls + li < len(data) else (voff(off + 8 + ls + li - 1) + 1)))
            off += 8 + ls + li
        n_ref = max([r[0] for r in recs] + [-1]) + 1
    else:
        off, n = 0, len(data)
        while off < n:
            e = data.find(b"\n", off)
            if e < 0:
                e = n
            line = data[off:e]
            if line and not line.startswith(b"#"):
                f = line.split(b"\t")
                chrom, pos0 = f[0].decode(), int(f[1]) - 1
                end = pos0 + max(1, len(f[3]))
                if len(f) > 7:
                    for kv in f[7].split(b";"):
                        if kv.startswith(b"END="):
                            try:
                                end = max(end, int(kv[4:]))
                            except ValueError:
                                pass
                if not names or names[-1] != chrom:
                    assert chrom not in names, "records are not grouped by contig"
                    names.append(chrom)
                recs.append((len(names) - 1, pos0, end, voff(off), voff(min(e + 1, n)) if e + 1 < n else voff(e) + 1))
            off = e + 1
        n_ref = len(names)

    def reg2bin(beg, end):
        end -= 1
        s, t = min_shift, ((1 << (depth * 3)) - 1) // 7
        for l in range(depth, 0, -1):
            if beg >> s == end >> s:
                return t + (beg >> s)
            s += 3
            t -= 1 << ((l - 1) * 3)
        return 0

    bins = [dict() for _ in range(n_ref)]
    linear = [dict() for _ in range(n_ref)]
    span = [[None, None, 0] for _ in range(n_ref)]
    for t, pos0, end, v0, v1 in recs:
        ch = bins[t].setdefault(reg2bin(pos0, end), [])
        if ch and ch[-1][1] == v0:
            ch[-1][1] = v1
        else:
            ch.append([v0, v1])
        for w in range(pos0 >> min_shift, ((end - 1) >> min_shift) + 1):
            linear[t].setdefault(w, v0)
        span[t][0] = v0 if span[t][0] is None else span[t][0]
        span[t][1] = v1
        span[t][2] += 1
    if is_bcf:
        aux = b""
    else:
        nm = b"".join(x.encode() + b"\0" for x in names)
        aux = struct.pack("<iiiiiii", 2, 1, 2, 0, ord("#"), 0, len(nm)) + nm
    out = bytearray(b"CSI\x01" + struct.pack("<iii", min_shift, depth, len(aux)) + aux + struct.pack("<i", n_ref))
    pseudo = ((1 << (depth * 3 + 3)) - 1) // 7 + 1
    for t in range(n_ref):
        n_win = (max(linear[t]) + 1) if linear[t] else 0
        lin = [linear[t].get(w, 0) for w in range(n_win)]
        for w in range(n_win - 2, -1, -1):
            if lin[w] == 0:
                lin[w] = lin[w + 1]

        def loff(b):  # the first window of the bin, in windows of the lowest level
            l, first = 0, 0
            while b >= first + (1 << (3 * l)):
                first += 1 << (3 * l)
                l += 1
            w = (b - first) << (3 * (depth - l))
            return lin[w] if w < n_win else 0

        out += struct.pack("<i", len(bins[t]) + (1 if span[t][2] else 0))
        for b in sorted(bins[t]):
            out += struct.pack("<IQi", b, loff(b), len(bins[t][b]))
            for v0, v1 in bins[t][b]:
                out += struct.pack("<QQ", v0, v1)
        if span[t][2]:
            out += struct.pack("<IQi", pseudo, 0, 2) + struct.pack("<QQQQ", span[t][0], span[t][1], span[t][2], 0)
    out += struct.pack("<Q", 0)
    csi_path = csi_path or path + ".csi"
    with open(csi_path, "wb") as fh:
        blob = bytes(out)
        for i in range(0, len(blob), 60000):
            fh.write(_bgzf_block(blob[i: i + 60000]))
        fh.write(_bgzf_block(b""))
    return csi_path
