"""Staged input of the read stage: which alignment records a batch of DNMs can look at.

The reference never reads an alignment file front to back: per DNM it fetches the records overlapping the
DNM position (read_collector.py:385-392; SVs: around both breakpoints, :478-497) and, with extended phasing
on, the records overlapping every het site of the DNM's window (:167), and asks for their mates (:400, :185).
`fetch_points` lists exactly those fetches for a batch, from the window lists of the site stage (uz_find);
libunfazed_io's selection (io_native.ReadsSource.select) then keeps the records they return plus their mates.
Only that selection is staged to the GPU, in the packed form (include/uz_types.h)."""
from __future__ import annotations

import numpy as np

from . import abi


def fetch_points(rcontig, start, dflags, site_pos, het_off, het_idx, params: abi.Params, vartype=None, end=None, cutoff=0.0,
                 allele_len=None):
    """-> (contig, lo, hi) int32 arrays: every fetch(contig, lo, hi) the read stage makes for the batch.
    With allele_len (per DNM: length of its longer allele) a fourth array comes back, `extra` (uint16): how many bases past the
    fetched position the read stage reads in a record -- the alleles at a DNM, nothing at a het site -- for the unit masks of
    ReadsSource.select.
    rcontig / start / dflags: per DNM (reads-table contig id, 0-based start, UZ_DF_*); het_off [n+1] / het_idx: the het
    lists of uz_find for the same batch; site_pos: positions of the sites table.  SV batches pass vartype / end and
    the kid's insert cutoff (collect_reads_sv fetches +-cutoff around both breakpoints)."""
    rcontig = np.asarray(rcontig, np.int64)
    start = np.asarray(start, np.int64)
    n = start.size
    fallback = (np.asarray(dflags) & abi.DF_FETCH_FALLBACK) != 0
    cs, los, his, exs = [], [], [], []
    al = np.minimum(np.asarray(allele_len if allele_len is not None else np.zeros(n)), 60000).astype(np.int64)
    if vartype is None or not np.any(np.asarray(vartype) != abi.VT_POINT):
        cs.append(rcontig)
        los.append(np.where(fallback, start, start - 1))
        his.append(start + 1)
        exs.append(al)
    else:
        vt = np.asarray(vartype)
        end = np.asarray(end, np.int64)
        icut = int(cutoff)
        pt = vt == abi.VT_POINT
        cs.append(rcontig[pt]); los.append(np.where(fallback[pt], start[pt], start[pt] - 1)); his.append(start[pt] + 1); exs.append(al[pt])
        for bp in (start[~pt], end[~pt]):
            cs.append(rcontig[~pt]); los.append(np.maximum(bp - icut, 0)); his.append(bp + icut); exs.append(np.zeros(int((~pt).sum()), np.int64))
    if not params.no_extended:
        het_off = np.asarray(het_off, np.int64)
        cnt = np.diff(het_off)
        hp = np.asarray(site_pos)[np.asarray(het_idx)[het_off[0]: het_off[-1]]].astype(np.int64)
        cs.append(np.repeat(rcontig, cnt)); los.append(hp); his.append(hp + 1); exs.append(np.zeros(hp.size, np.int64))
    c = np.concatenate(cs)
    lo = np.concatenate(los)
    hi = np.concatenate(his)
    keep = c >= 0
    if allele_len is not None:
        return c[keep].astype(np.int32), lo[keep].astype(np.int32), hi[keep].astype(np.int32), np.concatenate(exs)[keep].astype(np.uint16)
    return c[keep].astype(np.int32), lo[keep].astype(np.int32), hi[keep].astype(np.int32)
