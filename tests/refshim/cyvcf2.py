"""Stand-in for cyvcf2 0.31.0, used ONLY to import and run the reference
(/root/reference) in the authoring container when generating golden vectors
(tests/golden/make_golden.py).  Not part of the product; never shipped to or
used on the GPU box.  Duck-types exactly the attributes the reference touches
 over in-memory unfazed_amd.model.SiteRecord lists.
"""
import numpy as np

REGISTRY = {}  # path -> (samples, [SiteRecord])


def register(path, samples, records):
    REGISTRY[path] = (list(samples), list(records))


class _Info(dict):
    def get(self, key, default=None):
        return dict.get(self, key, default)


class Variant:
    def __init__(self, rec):
        self._r = rec
        self.CHROM = rec.chrom
        self.POS = rec.start + 1
        self.start = rec.start
        self.end = rec.end
        self.REF = rec.ref
        self.ALT = list(rec.alts)
        self.INFO = _Info(rec.info)
        # numpy arrays: the reference relies on numpy scalar division semantics
        # (x / 0.0 -> nan/inf with a warning, not ZeroDivisionError)
        self.gt_types = np.array(rec.gt_types, dtype=np.int32)
        self.gt_ref_depths = np.array(rec.ref_depths, dtype=np.int32)
        self.gt_alt_depths = np.array(rec.alt_depths, dtype=np.int32)
        self.gt_quals = np.array(rec.gt_quals, dtype=np.float32)
        self.genotypes = [list(g) for g in rec.genotypes] if rec.genotypes else None
        self.formats = {}

    def set_format(self, name, arr):
        self.formats[name] = np.array(arr)


class VCF:
    def __init__(self, path, *a, **k):
        if path not in REGISTRY:
            raise IOError("no such file: %s" % path)
        self.samples, self._records = REGISTRY[path]
        self._cursor = 0
        self.header_lines = []
        self.formats_added = []

    def __iter__(self):
        return self

    def __next__(self):
        if self._cursor >= len(self._records):
            raise StopIteration
        r = self._records[self._cursor]
        self._cursor += 1
        return Variant(r)

    def __call__(self, region):
        chrom, rng = region.rsplit(":", 1)
        # "c:-5-100" style negative starts: parity unpinned; clamp
        if rng.startswith("-"):
            b = "-" + rng[1:].split("-", 1)[0]
            e = rng[1:].split("-", 1)[1]
        else:
            b, e = rng.split("-", 1)
        beg, end = max(1, int(b)), int(e)
        for r in self._records:
            if r.chrom != chrom:
                continue
            # tabix: records overlapping the 1-based inclusive interval
            if r.start + 1 <= end and r.end >= beg:
                yield Variant(r)

    def add_to_header(self, line):
        self.header_lines.append(line)

    def add_format_to_header(self, d):
        self.formats_added.append(d)


class Writer:
    def __init__(self, path, tmpl, *a, **k):
        self.path = path
        self.tmpl = tmpl
        self.records = []

    def write_record(self, v):
        self.records.append(v)

    def close(self):
        pass
