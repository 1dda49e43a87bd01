"""BASELINE configs[2] / configs[4] as concrete synthetic inputs, shared by bench.py and tests/test_configs_gpu.py:
the sites table, the DNMs / events, the read clusters, the HBM-resident tables (the generator's HIP build fills them in place) and -- for the
staged pass -- what the decoders would leave in pinned memory per chunk of DNMs (site windows, records in the link form)."""
from __future__ import annotations

import numpy as np

from . import bigsynth
from .sites_np import CnvColumns, DnmColumns, breakpoint_dnms, make_clusters, make_sites, place_cnvs, place_dnms_full


def pinned_copy(pool, a):
    a = np.ascontiguousarray(a)
    out = pool.alloc(max(64, a.nbytes))[: a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


class BenchLoad:
    """workload: "snv" (configs[2]: SNV / INDEL DNMs placed uniformly), "snv_spaced" (configs[1]: one contig, DNMs at least `min_gap` apart),
    "cnv" (configs[4]: DEL / DUP events).  lo / hi: this rank's contiguous shard of the list (configs[3])."""

    def __init__(self, n_dnms, n_sites, workload="snv", seed_off=0, lo=None, hi=None, device=0, contig_lens=None, min_gap=12000, share=None):
        self.cnv = workload == "cnv"
        if share is not None:  # another load of the SAME list in this process (tests: the shards of configs[3] one after another): its tables are reused
            self.sc = sc = share.sc
            ev = share.ev_full
        else:
            self.sc = sc = make_sites(n_sites, seed=202) if contig_lens is None else make_sites(n_sites, seed=102, contig_lens=contig_lens)
        if share is not None:
            pass
        elif self.cnv:
            ev = place_cnvs(sc, n_dnms, seed=501 + seed_off, redraw_seed=502 + seed_off)
        elif workload == "snv_spaced":
            # stratified over the contig: neighbours at least half a stratum apart (40 Mb / 1 k DNMs: >= 20 kb), so no two read windows meet
            ev = place_dnms_full(sc, n_dnms, seed=101 + seed_off, uniform=False)
            gaps = np.diff(ev.start.astype(np.int64))[ev.contig[1:] == ev.contig[:-1]]
            if gaps.size and int(gaps.min()) < int(min_gap):
                raise ValueError("snv_spaced: DNMs %d bases apart (wanted >= %d): use a longer contig or fewer DNMs" % (int(gaps.min()), int(min_gap)))
        else:
            ev = place_dnms_full(sc, n_dnms, seed=201 + seed_off)
        self.n_total = ev.n
        self.ev_full = ev
        self.per_ev = 2 if self.cnv else 1
        self.cfg = bigsynth.make_cfg(seed=203 + seed_off)
        self.d_base = 0  # this rank's DNMs are [d_base, d_base + n) of the list the clusters were made over
        self.head_tlen = None
        if lo is not None and not self.cnv:
            # One rank's contiguous shard of a list every rank knows whole (configs[3]).  The pile-ups are those of the WHOLE list -- as
            # eight ranks reading one alignment file would see them: the clusters are made over the whole list and this rank generates the
            # ones its DNMs lie in (a cluster cut by a shard boundary is generated on both sides) -- and so is the head of the file the
            # insert cutoff is estimated from (read_collector.py:11-25): the first clusters, regenerated on the host.
            full = ev
            self.dn, self.cl = full, make_clusters(full)
            c0, c1 = self.cl.of_dnm(lo), self.cl.of_dnm(hi - 1) + 1
            ev = DnmColumns(ev.site_idx[lo:hi], ev.contig[lo:hi], ev.start[lo:hi], ev.end[lo:hi], ev.kind[lo:hi], ev.length[lo:hi],
                            ev.origin[lo:hi], ev.refs[lo:hi], ev.alts[lo:hi])
            self.d_base = lo
            self.wl = bigsynth.WorkloadOnGpu(self.cfg, sc, self.dn, self.cl, device=device, c0=c0, c1=c1)
            if c0 > 0:
                c_head = int(np.searchsorted(2 * self.cl.pair_off, 1000001, side="left"))
                _, arrs = bigsynth.reads_cpu(self.cfg, sc, self.dn, self.cl, 0, max(1, min(c_head, self.cl.n)), threads=8)
                self.head_tlen = np.asarray(arrs["tlen"][:1000001]).copy()
        else:
            if lo is not None:  # (events: their two breakpoints lie in different clusters -- the shard lays its own pile-ups)
                ev = CnvColumns(ev.contig[lo:hi], ev.start[lo:hi], ev.end[lo:hi], ev.vartype[lo:hi], ev.origin[lo:hi])
            # dn: the list the read generator lays its pile-ups around (the DNMs themselves / the breakpoints of the events)
            self.dn = breakpoint_dnms(ev) if self.cnv else ev
            self.cl = make_clusters(self.dn)
            self.wl = bigsynth.WorkloadOnGpu(self.cfg, sc, self.dn, self.cl, device=device)
        self.ev = ev
        self.n = ev.n
        self.ev_vt = ev.vartype if self.cnv else np.zeros(self.n, np.uint8)
        self.ev_refs = [b""] * self.n if self.cnv else ev.refs
        self.ev_alts = [b""] * self.n if self.cnv else ev.alts
        self.cutoff = None

    def adopt(self, eng, P):
        """the generator's tables handed to the library where they lie (HBM) -> (sites id, family id, reads id)"""
        from unfazed_amd.hostpath import concordant_cutoff
        sid = eng.adopt_sites(self.wl.sites_view())
        fid = eng.adopt_family(sid, self.wl.family_view())
        rid = eng.adopt_reads(self.wl.reads_view())
        # concordant insert cutoff: host scalar per kid (read_collector.py:11-25) from the first records
        self.cutoff = concordant_cutoff(self.wl.tlen_head() if self.head_tlen is None else self.head_tlen, P.readlen, 3)
        return sid, fid, rid

    def view_of(self, a, b):
        from unfazed_amd import abi
        ev = self.ev
        return abi.dnms_view(ev.contig[a:b], ev.contig[a:b], ev.start[a:b], ev.end[a:b], self.ev_vt[a:b], self.ev_refs[a:b], self.ev_alts[a:b], self.cutoff)

    def window_sites(self, a, b, P):
        """indices of the sites inside the windows of DNMs [a, b) (the reference queries the indexed VCF per DNM region,
        informative_site_finder.py:399-420: no site outside a window is ever decoded)"""
        sc, ev = self.sc, self.ev
        sd = int(P.search_dist) + 2
        co_s = np.asarray(sc.contig_off, np.int64)
        kk = np.zeros(sc.n + 1, np.int32)
        for c in np.unique(ev.contig[a:b]):
            if c < 0:
                continue
            m = ev.contig[a:b] == c
            pc = sc.pos[co_s[c]: co_s[c + 1]]
            lo_i = np.searchsorted(pc, ev.start[a:b][m].astype(np.int64) - sd, "left") + co_s[c]
            hi_i = np.searchsorted(pc, ev.end[a:b][m].astype(np.int64) + sd, "right") + co_s[c]
            np.add.at(kk, lo_i, 1)
            np.add.at(kk, hi_i, -1)
        return np.nonzero(np.cumsum(kk[:-1]) > 0)[0]

    def pinned_sites(self, pool, sel, eight_bit=True):
        """the site + genotype columns of `sel` in pinned memory, as one slab -> (held view, site columns, genotype columns, wide list, n)"""
        from unfazed_amd import abi
        sc = self.sc
        co_s = np.asarray(sc.contig_off, np.int64)
        pool.new_slab(int(sel.size) * 28 + (1 << 20))
        hs_k = {k: pinned_copy(pool, getattr(sc, k)[sel]) for k in ("pos", "sflags", "ref_base", "alt_base", "gt")}
        hs_k["contig_off"] = pinned_copy(pool, np.searchsorted(sel, co_s).astype(np.int64))
        hg_k = {k: [getattr(sc, k)[m][sel] for m in range(3)] for k in ("rd", "ad", "gq")}
        wide_k = None
        if eight_bit:  # the nine genotype columns in eight bits (uz_types.h: depths of 255 and more, or missing, through the wide list)
            r8, a8, g8, wide_k = abi.family_columns8(hg_k["rd"], hg_k["ad"], hg_k["gq"])
            hg_k = dict(rd=list(r8), ad=list(a8), gq=list(g8))
        hg_k = {k: [pinned_copy(pool, x) for x in v] for k, v in hg_k.items()}
        svk = abi.SitesView()
        svk.n_sites, svk.n_contigs = int(sel.size), len(sc.contig_off) - 1
        for k in ("contig_off", "pos", "sflags", "ref_base", "alt_base"):
            setattr(svk, k, hs_k[k].ctypes.data)
        pool.end_slab()
        return abi.Held(svk, hs_k), hs_k, hg_k, wide_k, int(sel.size)

    def stage(self, eng, P, mode, fid, pool, chunks=None, last_chunk=0.7, first_chunk=None, sites16=False, per_chunk_sites=True, log_first=False):
        """What the decoders would leave in pinned memory for a staged pass: per chunk of DNMs (whole clusters) the records the chunk's fetches
        return + their mates, in the link form (selected on the host out of the generator's table), and the site windows.
        -> (chunks for pipeline.run_pipelined, stats)"""
        from unfazed_amd import io_native, shard
        from unfazed_amd.staging import fetch_points
        ev, sc, cl, wl, n = self.ev, self.sc, self.cl, self.wl, self.n
        dv = self.view_of(0, n)
        co, ci, cf, ho, hi = eng.find(fid, dv, P, mode)
        out, staged_bytes, staged_records = [], 0, 0
        slab_hint = 768 << 20
        # chunks of events (their records: the clusters of their generator entries); the last one smaller: its read stage is the only one
        # nothing hides -- the link is idle by then
        ecuts = shard.chunk_plan(n, chunks or None, last_chunk, first_chunk=first_chunk)
        for k in range(len(ecuts) - 1):
            a, b = ecuts[k], ecuts[k + 1]
            if b <= a:
                continue
            c0, c1 = cl.of_dnm(self.d_base + self.per_ev * a), cl.of_dnm(self.d_base + self.per_ev * b - 1) + 1
            part_full = wl.download(c0, c1)
            src = io_native.ReadsSource(part_full)
            alen = np.array([max(len(r), len(x)) for r, x in zip(self.ev_refs[a:b], self.ev_alts[a:b])], np.int64)
            fc, flo, fhi, fex = fetch_points(ev.contig[a:b], ev.start[a:b], np.zeros(b - a, np.uint8), sc.pos, ho[a: b + 1], hi, P,
                                             vartype=self.ev_vt[a:b], end=ev.end[a:b], cutoff=self.cutoff, allele_len=alen)
            # qualities as counts + short lists, and of every record's rows only the 32-base units that hold a fetched position; SV batches
            # the same way: their +-cutoff fetches stage no base unit -- collect_reads_sv reads none -- and the read stage asks for quality
            # bits of records that pass goodread only, which the lists hold (uz_types.h).  The chunk's columns back to back in one pinned
            # block: they cross the link as one copy.
            pool.new_slab(slab_hint)
            part = src.select(fc, flo, fhi, alloc=pool.alloc, all_bases=bool(P.no_extended), lists=True, extra=fex, wide_no_units=self.cnv)
            slab_hint = max(64 << 20, int(pool.end_slab() * 1.3))
            if log_first and not out:  # development aid: the first chunk, column by column
                import sys
                nrec = int(part.view.n_segs)
                print("[link bytes] %d records for %d DNMs:" % (nrec, b - a), {k: (int(x.nbytes), round(x.nbytes / max(1, b - a), 1)) for k, x in part.arrays.items()},
                      file=sys.stderr)
            del src, part_full
            out.append(dict(a=a, b=b, records=part, dnms=self.view_of(a, b), sites=None))
            staged_records += int(part.view.n_segs)
            staged_bytes += link_bytes(part.view)
        n_sites_staged, site_bytes = 0, 0
        if per_chunk_sites:
            # the site columns cut per chunk (the windows of the chunk's DNMs): the site stage of chunk k + 1 then runs while the records of
            # chunk k are still on the link (config 5 the same way: the allele-balance stage of a chunk runs on the chunk's own windows)
            for c in out:
                held, hs_k, hg_k, wide_k, ns = self.pinned_sites(pool, self.window_sites(c["a"], c["b"], P), eight_bit=not sites16)
                c["sites"] = (held, hs_k, hg_k, wide_k)
                n_sites_staged += ns
                site_bytes += ns * (4 + 1 + 1 + 1 + 1 + (18 if sites16 else 9)) + (0 if wide_k is None else len(wide_k[0]) * 32)
        return out, dict(read_bytes=staged_bytes, records=staged_records, sites=n_sites_staged, site_bytes=site_bytes)

    def free(self):
        self.wl.free()


def link_bytes(pv) -> int:
    """bytes of one table in the link form (uz_reads_packed_view), column by column"""
    dic = bool(pv.tup) or bool(pv.tup8)
    lists_f = bool(pv.n_low) or bool(dic and pv.tup_n_low)
    fixed = ((2 if pv.pair_d8 else 5 if pv.mate_d8 else 7) if pv.start_d8 else 8 if pv.start_d else 16) + (4 if pv.end else 0) + (2 if pv.umask else 0) + (1 if pv.tup8 else 2 if pv.tup else 8)  # start, tlen, mate, qname | end | umask | tup8 / tup or flag .. aux
    t8 = (512 + 2 * int(pv.n_tup_esc) + 4 * ((int(pv.n_segs) + 1023) // 1024 + 1)) if pv.tup8 else 0  # hot table, escape list, span offsets
    return (int(pv.n_segs) * fixed + t8 + int(pv.n_tup) * 11 + int(pv.n_esc16) * 12 + int(pv.n_cigar_total) * 4
            + ((0 if dic else int(pv.n_segs)) + int(pv.n_qlow_pos) * (2 if pv.qlow_pos_wide else 1) if lists_f else int(pv.n_row_units) * 4)
            + (int(pv.n_seq_units) * 8 + int(pv.n_exc) * 7 if pv.seq2 else int(pv.n_seq_units) * 16)
            + (int(pv.n_bl) * (2 if pv.bl_wide else 1) + (int(pv.n_bl) + 3) // 4 + (int(pv.n_tup) if pv.tup_n_bl else int(pv.n_segs) if pv.bl_n else 0))
            + ((int(pv.n_pk_spans) + 1) * 88 if pv.pk_sums else 0))  # (the packer's span sums: uz_types.h pk_sums)
