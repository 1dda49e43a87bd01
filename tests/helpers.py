"""Shared helpers of the parity tests: build the decoded tables of a synth.small
dataset and drive a backend through the host path."""
import contextlib
import copy
import io

import numpy as np

from synth.small import SmallConfig, make_small
from unfazed_amd import abi
from unfazed_amd.hostpath import PhasingHost
from unfazed_amd.model import ReadsTable, SitesTable

RUN_DEFAULTS = dict(
    threads=1, build="38", no_extended=False, multithread_proc_min=1000, quiet_mode=False,
    ab_homref=[0.0, 0.2], ab_homalt=[0.8, 1.0], ab_het=[0.2, 0.8], min_gt_qual=20, min_depth=10,
    search_dist=5000, insert_size_max_sample=1000000, stdevs=3, min_map_qual=1, readlen=151,
    split_error_margin=5,
)


def split_kwargs(kw):
    cfgkw = {k: v for k, v in kw.items() if k in SmallConfig.__dataclass_fields__}
    runkw = {k: v for k, v in kw.items() if k not in cfgkw}
    return cfgkw, runkw


def tables(ds):
    sites = SitesTable.from_records(ds.sites, ds.samples)
    reads = {}
    for kid, segs in ds.reads.items():
        rt = ReadsTable.from_segments(segs, ds.contigs)
        rt.tlen_head = np.array([s.tlen for s in segs], dtype=np.int32)
        reads["mem://%s.bam" % kid] = rt
    return sites, reads


def params_from(a):
    return abi.make_params(
        search_dist=a["search_dist"], min_gt_qual=a["min_gt_qual"], min_depth=a["min_depth"],
        min_map_qual=a["min_map_qual"], readlen=a["readlen"], no_extended=a["no_extended"],
        insert_size_max_sample=a["insert_size_max_sample"], ab_homref=a["ab_homref"], ab_homalt=a["ab_homalt"],
        ab_het=a["ab_het"],
    )


def run_host(backend, ds, sites=None, reads=None, **runkw):
    """-> (records, annotated dnms, stderr text)"""
    a = dict(RUN_DEFAULTS)
    a.update(runkw)
    if sites is None:
        sites, reads = tables(ds)
    host = PhasingHost(backend, sites, reads)
    dn = copy.deepcopy(ds.dnms)
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        recs = host.run_read_phasing(
            dn, ds.pedigrees, a["threads"], a["build"], a["no_extended"], a["multithread_proc_min"],
            a["quiet_mode"], params_from(a), a["search_dist"], a["insert_size_max_sample"], a["stdevs"], a["readlen"],
        )
    return recs, dn, err.getvalue()


def norm_records(recs):
    out = {}
    for k, r in recs.items():
        out[k] = {kk: (sorted(v) if isinstance(v, list) else v) for kk, v in r.items()}
    return out


def dnm_sites(dnms):
    return [(d["chrom"], d["start"], d["end"], d["kid"], d.get("candidate_sites"), d.get("het_sites")) for d in dnms]


# ---------------------------------------------------------------- compact golden form (tests/golden/make_golden_wide.py)
def _sha(obj):
    import hashlib
    import json
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()[:24]


def compact_records(recs):
    """Records with the read-name lists replaced by (count, digest of the sorted names); everything else kept."""
    out = {}
    for k, r in recs.items():
        c = {}
        for kk, v in r.items():
            if kk in ("dad_reads", "mom_reads"):
                c[kk] = {"n": len(v), "sha": _sha(sorted(v))}
            else:
                c[kk] = sorted(v) if isinstance(v, list) else v
        out[k] = c
    return out


def compact_dnms(dnms):
    """Per returned DNM: identity + counts and digests of the annotated site lists (order included)."""
    out = []
    for d in dnms:
        cs, hs = d.get("candidate_sites"), d.get("het_sites")
        out.append(dict(chrom=d["chrom"], start=d["start"], end=d["end"], kid=d["kid"],
                        cand=None if cs is None else {"n": len(cs), "sha": _sha(cs)},
                        het=None if hs is None else {"n": len(hs), "sha": _sha(hs)}))
    return out


def reverse_ties(ds):
    """The same records with the order of records sharing (contig, position) reversed: the reference's chaining is
    first-come (connect_reads, read_collector.py:76-152), so the fetch order of ties is part of the input."""
    for kid in ds.reads:
        segs = list(reversed(ds.reads[kid]))
        segs.sort(key=lambda s: (s.tid if s.tid >= 0 else 1 << 30, s.pos))
        ds.reads[kid] = segs
    return ds
