"""Dumps what an independent, minimal reader makes of the htslib-written index files the reference ships (test/data/*.bai, *.tbi):
per reference the number of bins (37450 pseudo-bin apart), chunks, linear entries and three sums.  tests/test_index_refdata.py holds
the native readers (csrc/io_index.hpp) against this dump.  Run in the authoring container: python tests/golden/make_index_golden.py"""
import gzip
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
M = (1 << 62) - 1


def refs_summary(d, off, n_ref):
    out = []
    for _ in range(n_ref):
        (n_bin,) = struct.unpack_from("<i", d, off); off += 4
        nb = nc = sb = se = 0
        pseudo = None
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, off); off += 8
            chunks = struct.unpack_from("<%dQ" % (2 * n_chunk), d, off); off += 16 * n_chunk
            if b == 37450:
                pseudo = list(chunks)
                continue
            nb += 1; nc += n_chunk; sb += sum(chunks[0::2]); se += sum(chunks[1::2])
        (n_intv,) = struct.unpack_from("<i", d, off); off += 4
        lin = struct.unpack_from("<%dQ" % n_intv, d, off); off += 8 * n_intv
        out.append(dict(bins=nb, chunks=nc, linear=n_intv, sum_beg=sb & M, sum_end=se & M, sum_linear=sum(lin) & M, pseudo_bin=pseudo))
    return out, off


def dump(path):
    raw = open(path, "rb").read()
    if path.endswith(".tbi"):
        d = gzip.decompress(raw)
        assert d[:4] == b"TBI\x01"
        n_ref, fmt, cs, cb, ce, meta, skip, l_nm = struct.unpack_from("<8i", d, 4)
        names = d[36: 36 + l_nm].split(b"\0")[:-1]
        refs, _ = refs_summary(d, 36 + l_nm, n_ref)
        return dict(kind="tbi", n_ref=n_ref, format=fmt, names=[x.decode() for x in names], refs=refs)
    assert raw[:4] == b"BAI\x01"
    (n_ref,) = struct.unpack_from("<i", raw, 4)
    refs, off = refs_summary(raw, 8, n_ref)
    return dict(kind="bai", n_ref=n_ref, refs=refs, trailing_bytes=len(raw) - off)


if __name__ == "__main__":
    rd = os.path.join(HERE, "refdata")
    out = {f: dump(os.path.join(rd, f)) for f in sorted(os.listdir(rd)) if f.endswith((".bai", ".tbi"))}
    json.dump(out, open(os.path.join(HERE, "index_refdata.json"), "w"), indent=0)
    print({k: (v["n_ref"], sum(r["chunks"] for r in v["refs"])) for k, v in out.items()})
