"""Stand-in for pysam 0.22.1, used ONLY to import and run the reference
(/root/reference) in the authoring container when generating golden vectors.
Not part of the product.  Duck-types the attributes the reference touches
 over in-memory unfazed_amd.model.Segment lists.
"""
import bisect

REGISTRY_INDEX = {}
REGISTRY = {}  # path -> (contig names, [Segment] in file order)
COUNTERS = {"fetch": 0, "mate": 0, "open": 0}


def register(path, contigs, segments):
    REGISTRY[path] = (list(contigs), list(segments))
    REGISTRY_INDEX.pop(path, None)


class AlignedSegment:
    __slots__ = ("_s", "_idx", "_refpos")

    def __init__(self, seg, idx):
        self._s = seg
        self._idx = idx
        self._refpos = None

    # flags
    @property
    def flag(self):
        return self._s.flag

    @property
    def is_qcfail(self):
        return bool(self._s.flag & 512)

    @property
    def is_unmapped(self):
        return bool(self._s.flag & 4)

    @property
    def is_duplicate(self):
        return bool(self._s.flag & 1024)

    @property
    def is_secondary(self):
        return bool(self._s.flag & 256)

    @property
    def is_supplementary(self):
        return bool(self._s.flag & 2048)

    @property
    def mate_is_unmapped(self):
        return bool(self._s.flag & 8)

    @property
    def mapping_quality(self):
        return self._s.mapq

    @property
    def reference_id(self):
        return self._s.tid

    @property
    def next_reference_id(self):
        return self._s.mtid

    @property
    def query_qualities(self):
        return None if self._s.qual is None else list(self._s.qual)

    @property
    def cigartuples(self):
        return list(self._s.cigar) if self._s.cigar else None

    @property
    def tlen(self):
        return self._s.tlen

    @property
    def query_name(self):
        return self._s.qname

    @property
    def query_sequence(self):
        return self._s.seq if self._s.seq else None

    @property
    def reference_start(self):
        return self._s.pos

    @property
    def reference_end(self):
        if (self._s.flag & 4) or not self._s.cigar:
            return None
        return self._s.pos + self._s.ref_len

    def has_tag(self, tag):
        return tag == "SA" and self._s.has_sa

    def get_reference_positions(self, full_length=False):
        if not self._s.cigar:
            return []
        out = []
        pos = self._s.pos
        for op, l in self._s.cigar:
            if op in (4, 1):  # S, I
                if full_length:
                    out.extend([None] * l)
            elif op in (0, 7, 8):  # M, =, X
                out.extend(range(pos, pos + l))
                pos += l
            elif op in (2, 3):  # D, N
                pos += l
        return out


class AlignmentFile:
    def __init__(self, path, mode="rb", reference_filename=None):
        if path not in REGISTRY:
            raise IOError("no such file: %s" % path)
        self.contigs, self._segs = REGISTRY[path]
        self._tid = {c: i for i, c in enumerate(self.contigs)}
        COUNTERS["open"] += 1
        idx = REGISTRY_INDEX.get(path)
        if idx is None:
            # per-contig (indices, starts, max span) for O(log n) fetch
            idx = {}
            for i, s in enumerate(self._segs):
                d = idx.setdefault(s.tid, [[], [], 0])
                d[0].append(i)
                d[1].append(s.pos)
                d[2] = max(d[2], s.endpos - s.pos)
            REGISTRY_INDEX[path] = idx
        self._index = idx

    def __iter__(self):
        for i, s in enumerate(self._segs):
            yield AlignedSegment(s, i)

    def _fetch_tid(self, tid, start, stop):
        if tid not in self._index:
            return
        ids, starts, span = self._index[tid]
        k = bisect.bisect_left(starts, start - span)
        while k < len(ids) and starts[k] < stop:
            s = self._segs[ids[k]]
            if s.endpos > start:
                yield AlignedSegment(s, ids[k])
            k += 1

    def fetch(self, contig=None, start=None, stop=None, tid=None, multiple_iterators=False):
        COUNTERS["fetch"] += 1
        if tid is None:
            if contig not in self._tid:
                raise ValueError("invalid contig `%s`" % contig)
            tid = self._tid[contig]
        return self._fetch_tid(tid, int(start), int(stop))

    def mate(self, read):
        COUNTERS["mate"] += 1
        s = read._s
        if not (s.flag & 1):
            raise ValueError("read %s: is unpaired" % s.qname)
        if s.flag & 8:
            raise ValueError("mate %s: is unmapped" % s.qname)
        want = (s.flag ^ 192) & 192
        if s.mtid < 0:
            raise ValueError("mate not found")
        for m in self._fetch_tid(s.mtid, s.mpos, s.mpos + 1):
            if (m._s.flag & want) != 0 and m._s.qname == s.qname:
                return m
        raise ValueError("mate not found")
