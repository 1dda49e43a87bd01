"""Host side of the staged record format (libunfazed_io: uz_reads_pack, uz_reads_select_*; staging.fetch_points):
 * the packed columns say what the ASCII columns say (independent numpy unpacking);
 * a selection holds exactly the records the fetches return plus the closure under `mate`, in order, with mate
   links renumbered;
 * the read stage sees nothing else: the oracle on the selected records gives, per DNM, what it gives on the whole
   table (benchmark-scale generator, DNM chunks cut inside clusters)."""
import numpy as np
import pytest

from oracle import oracle as orc
from synth import bigsynth
from synth.sites_np import make_clusters, make_sites, place_dnms_full
from unfazed_amd import abi, io_native
from unfazed_amd.hostpath import concordant_cutoff
from unfazed_amd.staging import fetch_points

LUT = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)


def _workload(n_dnms=260, seed=7):
    sc = make_sites(40_000, seed=seed, contig_lens=[6e6, 4e6, 2e6])
    dn = place_dnms_full(sc, n_dnms, seed=seed + 1, indel_frac=0.2)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=seed + 2)
    cfg.n_clusters = cl.n
    rh, arrs = bigsynth.reads_cpu(cfg, sc, dn, cl, 0, cl.n, threads=4)
    # the generator writes A/C/G/T only: sprinkle the other BAM codes (the two-bit rows list them apart)
    rng = np.random.default_rng(seed + 3)
    n = int(rh.view.n_segs)
    for i in rng.integers(0, n, max(4, n // 40)):
        ls, r0 = int(arrs["l_seq"][i]), int(arrs["sq_off16"][i]) * 16
        for k in rng.integers(0, ls, int(rng.integers(1, 4))):
            arrs["seq"][r0 + k] = LUT[int(rng.choice([15, 15, 15, 5, 0, 14]))]
    return sc, dn, cl, rh, arrs


def _sites_views(sc):
    sv = abi.SitesView()
    keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=sc.pos, sflags=sc.sflags,
                ref_base=sc.ref_base, alt_base=sc.alt_base)
    sv.n_sites, sv.n_contigs = sc.n, len(sc.contig_off) - 1
    for k, a in keep.items():
        setattr(sv, k, a.ctypes.data)
    return abi.Held(sv, keep), abi.family_view(sc.gt, sc.rd, sc.ad, sc.gq)


def _unpack_row(pk, uoff, i, ls, soff=None):
    """bases (None for a record staged without them) and quality bits of record i; uoff / soff: quality-plane / seq4 row
    offsets in units (the same when every record carries its bases)"""
    u = int(abi.row_units(ls))
    soff = uoff if soff is None else soff
    seq = None
    if not (pk.arrays["aux"][i] & abi.AUX_NO_SEQ):
        if "seq4" in pk.arrays:
            row = pk.arrays["seq4"][soff[i] * 16: (soff[i] + u) * 16]
            seq = LUT[np.stack([row >> 4, row & 15], 1).ravel()[:ls]]
        else:  # two-bit rows (first base in the top bits of a byte) + the listed bases
            row = pk.arrays["seq2"][soff[i] * 8: (soff[i] + u) * 8]
            codes = np.stack([row >> 6, (row >> 4) & 3, (row >> 2) & 3, row & 3], 1).ravel()[:ls]
            seq = LUT[1 << codes].copy()
            ne = int(pk.view.n_exc)
            er = pk.arrays["exc_rec"][:ne]
            a, b = np.searchsorted(er, i, "left"), np.searchsorted(er, i, "right")
            assert (codes[pk.arrays["exc_pos"][a:b]] == 0).all()
            seq[pk.arrays["exc_pos"][a:b]] = LUT[pk.arrays["exc_code"][a:b]]
    if "qlow" in pk.arrays:
        low = np.unpackbits(pk.arrays["qlow"][uoff[i] * 4: (uoff[i] + u) * 4], bitorder="little")[:ls]
    else:  # list form: the count of every record, the positions of the records with bases and at most QLOW_LIST_MAX of them
        n = int(pk.view.n_segs)
        nl = pk.arrays["n_low"][:n].astype(np.int64)
        listed = ((pk.arrays["aux"][:n] & abi.AUX_NO_SEQ) == 0) & (nl <= abi.QLOW_LIST_MAX)
        loff = np.concatenate([[0], np.cumsum(np.where(listed, nl, 0))])
        assert loff[-1] == pk.view.n_qlow_pos and not pk.view.qlow_pos_wide
        low = (int(nl[i]), None)
        if listed[i]:
            bits = np.zeros(ls, np.uint8)
            bits[pk.arrays["qlow_pos"][loff[i]: loff[i + 1]]] = 1
            assert bits.sum() == nl[i]
            low = (int(nl[i]), bits)
    return seq, low


def _same_low(low, bits):
    """quality bits of a record in either form against the expected 0/1 vector"""
    if isinstance(low, tuple):
        cnt, got = low
        assert cnt == min(int(bits.sum()), 255)
        assert got is None or np.array_equal(got, bits)
        return True
    return np.array_equal(low, bits)


@pytest.mark.parametrize("two_bit", [True, False])
def test_pack_matches_ascii_columns(two_bit):
    sc, dn, cl, rh, arrs = _workload(60)
    for thr in (20, 13, 0, 300):
        pk = io_native.pack_reads(rh, thr, two_bit=two_bit, lists=two_bit, with_end=None if two_bit else True)
        assert ("end" in pk.arrays) == (not two_bit)  # the generator's `end` is what the CIGAR gives: the column can stay home
        if two_bit:
            n_other = int((~np.isin(arrs["seq"], LUT[[1, 2, 4, 8]])).sum())  # (rows are written back to back: every byte is a base)
            assert pk.view.n_exc >= 4 and pk.view.n_exc <= n_other and np.all(np.diff(pk.arrays["exc_rec"][: pk.view.n_exc].astype(np.int64)) >= 0)
        n = int(rh.view.n_segs)
        assert pk.view.n_segs == n and pk.view.min_base_qual == thr
        for name, _ in abi.PACKED_RECORD_COLS:
            if name in pk.arrays:
                assert np.array_equal(pk.arrays[name][:n], arrs[name][:n]), name
        uoff = np.concatenate([[0], np.cumsum(abi.row_units(arrs["l_seq"][:n]))])
        coff = np.concatenate([[0], np.cumsum(arrs["n_cigar"][:n].astype(np.int64))])
        assert pk.view.n_row_units == uoff[-1] and pk.view.n_cigar_total == coff[-1]
        rng = np.random.default_rng(thr)
        for i in rng.integers(0, n, 300):
            ls = int(arrs["l_seq"][i])
            r0 = int(arrs["sq_off16"][i]) * 16
            seq, low = _unpack_row(pk, uoff, i, ls)
            assert np.array_equal(seq, arrs["seq"][r0: r0 + ls])
            assert _same_low(low, (arrs["qual"][r0: r0 + ls].astype(np.int64) < min(max(thr, 0), 256)).astype(np.uint8))
            c0 = int(arrs["cigar_off"][i])
            assert np.array_equal(pk.arrays["cigar"][coff[i]: coff[i + 1]], arrs["cigar"][c0: c0 + int(arrs["n_cigar"][i])])


def test_pack_rejects_foreign_characters():
    sc, dn, cl, rh, arrs = _workload(20)
    arrs["seq"][5] = ord("a")
    try:
        io_native.pack_reads(rh, 20)
    except io_native.IoError as e:
        assert "alphabet" in str(e)
    else:
        raise AssertionError("a lowercase base must be refused, not folded")


def _subset_ascii(arrs, idx, n_contigs):
    """ASCII view of the records idx (ascending) of the table `arrs`, mates renumbered"""
    new = np.full(arrs["start"].shape[0], -1, np.int64)
    new[idx] = np.arange(idx.size)
    out = {}
    for k in ("start", "end", "flag", "mapq", "aux", "tlen", "qname", "cigar_off", "n_cigar", "l_seq", "sq_off16"):
        out[k] = np.ascontiguousarray(arrs[k][idx])
    mt = arrs["mate"][idx]
    out["mate"] = np.where(mt >= 0, new[np.maximum(mt, 0)], -1).astype(np.int32)
    out["cigar"], out["seq"], out["qual"] = arrs["cigar"], arrs["seq"], arrs["qual"]  # rows stay where they are
    co = np.searchsorted(idx, arrs["contig_off"]).astype(np.int64)
    out["contig_off"] = co
    span = np.zeros(n_contigs, np.int32)
    for c in range(n_contigs):
        if co[c + 1] > co[c]:
            span[c] = (out["end"][co[c]: co[c + 1]] - out["start"][co[c]: co[c + 1]]).max()
    out["max_span"] = span
    v = abi.ReadsView()
    v.n_segs, v.n_contigs = idx.size, n_contigs
    for k, a in out.items():
        setattr(v, k, a.ctypes.data)
    v.n_cigar_total, v.n_sq_bytes = arrs["cigar"].shape[0], arrs["seq"].shape[0]
    v.n_qnames = int(arrs["qname"].max()) + 1
    return abi.Held(v, out)


@pytest.mark.parametrize("two_bit,src_lists,out_lists", [(True, True, True), (False, False, False), (True, False, True)])
def test_selection_is_what_the_fetches_return_and_the_oracle_needs_nothing_else(two_bit, src_lists, out_lists):
    sc, dn, cl, rh, arrs = _workload(260)
    n, nc = dn.n, len(sc.contig_off) - 1
    P = abi.make_params()
    sh, fh = _sites_views(sc)
    cutoff = concordant_cutoff(arrs["tlen"], P.readlen, 3)
    dv = abi.dnms_view(dn.contig, dn.contig, dn.start, dn.end, np.zeros(n, np.uint8), dn.refs, dn.alts, cutoff)
    found = orc.find(P, sh, fh, dv, abi.FIND_SECOND_WINDOW)
    co, ci, cf, ho, hi = found
    want = orc.phase(P, sh, rh, dv, found, keep_lists=True)
    assert (want["status"] == abi.ST_OK).sum() > 30
    pk = io_native.pack_reads(rh, P.min_gt_qual, two_bit=two_bit, lists=src_lists, with_end=True)  # (a source of selections keeps `end`)
    src = io_native.ReadsSource(pk)
    N = int(rh.view.n_segs)
    contig_of_rec = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
    bounds = [0, 37, 111, 200, n]  # inside clusters
    total = 0
    for a, b in zip(bounds[:-1], bounds[1:]):
        fc, flo, fhi = fetch_points(dn.contig[a:b], dn.start[a:b], np.zeros(b - a, np.uint8), sc.pos, ho[a: b + 1], hi, P)
        part, idx = src.select(fc, flo, fhi, want_index=True, lists=out_lists, with_end=None if out_lists else True)
        assert ("tup" in part.arrays or "tup8" in part.arrays) and part.view.n_tup < 2000  # the small columns travel as a dictionary
        part.arrays.update(abi.small_columns(part)); part.arrays.update(abi.wide_columns(part))
        # brute force: overlap of any fetch, then the closure under mate
        keep = np.zeros(N, bool)
        for c, lo, h in zip(fc, flo, fhi):
            keep |= (contig_of_rec == c) & (arrs["start"][:N] < h) & (arrs["end"][:N] > lo)
        direct = keep.copy()
        for _ in range(4):
            m = arrs["mate"][:N][keep]
            keep[m[m >= 0]] = True
        assert np.array_equal(np.nonzero(keep)[0], idx)
        total += idx.size
        # packed columns of the selection = the packed columns of those records; a record no fetch returns (reachable
        # only as a mate) travels without its bases
        assert ("end" in part.arrays) == (not out_lists)  # (left out by default; the plane-form case below asks for it)
        for name, _ in abi.PACKED_RECORD_COLS:
            if name not in ("mate", "aux", "qname") and name in part.arrays:
                assert np.array_equal(part.arrays[name][: idx.size], pk.arrays[name][idx]), name
        # the pair form numbers the names of the selection by first appearance; qname_map leads back to the source's ids
        assert "pair_d8" in part.arrays and part.view.n_qnames == part.qname_map.size and np.all(np.diff(part.qname_map.astype(np.int64)) > 0)
        assert np.array_equal(part.qname_map[part.arrays["qname"][: idx.size]], pk.arrays["qname"][idx])
        no_seq = (part.arrays["aux"][: idx.size] & abi.AUX_NO_SEQ) != 0
        assert np.array_equal(no_seq, ~direct[idx]) and 0.3 < no_seq.mean() < 0.6
        SIMPLE = np.uint8(48)  # cigar_compact: a record that is one M / = / X over the read names the operation in its aux byte
        assert np.array_equal(part.arrays["aux"][: idx.size] & ~np.uint8(abi.AUX_NO_SEQ) & ~SIMPLE, pk.arrays["aux"][idx])
        code = (part.arrays["aux"][: idx.size] & SIMPLE) >> 4
        assert part.view.cigar_compact == 1 and part.view.n_cigar_omitted == int((code != 0).sum()) > 0.9 * idx.size
        coff_f = np.concatenate([[0], np.cumsum(pk.arrays["n_cigar"][:N].astype(np.int64))])
        coff_p = np.concatenate([[0], np.cumsum(np.where(code != 0, 0, part.arrays["n_cigar"][: idx.size].astype(np.int64)))])
        assert coff_p[-1] == part.view.n_cigar_total
        for k in np.random.default_rng(a + 1).integers(0, idx.size, 200):
            words = pk.arrays["cigar"][coff_f[idx[k]]: coff_f[idx[k] + 1]]
            if code[k]:
                assert words.size == 1 and words[0] == (int(part.arrays["l_seq"][k]) << 4 | {1: 0, 2: 7, 3: 8}[int(code[k])])
            else:
                assert np.array_equal(part.arrays["cigar"][coff_p[k]: coff_p[k + 1]], words)
        assert part.view.n_seq_units == int(abi.row_units(part.arrays["l_seq"][: idx.size])[~no_seq].sum())
        everything, _ = src.select(fc, flo, fhi, want_index=True, all_bases=True, lists=out_lists)
        everything.arrays.update(abi.small_columns(everything)); everything.arrays.update(abi.wide_columns(everything))
        assert everything.view.n_seq_units == everything.view.n_row_units and not (everything.arrays["aux"][: idx.size] & abi.AUX_NO_SEQ).any()
        mt = part.arrays["mate"][: idx.size]
        assert np.array_equal(idx[mt[mt >= 0]], pk.arrays["mate"][idx][mt >= 0])
        uoff_f = np.concatenate([[0], np.cumsum(abi.row_units(pk.arrays["l_seq"][:N]))])
        units_p = abi.row_units(part.arrays["l_seq"][: idx.size])
        uoff_p = np.concatenate([[0], np.cumsum(units_p)])
        soff_p = np.concatenate([[0], np.cumsum(np.where(no_seq, 0, units_p))])
        for k in np.random.default_rng(a).integers(0, idx.size, 80):
            ls = int(part.arrays["l_seq"][k])
            s1, q1 = _unpack_row(part, uoff_p, k, ls, soff_p)
            s2, q2 = _unpack_row(pk, uoff_f, idx[k], ls)
            r0 = int(arrs["sq_off16"][idx[k]]) * 16
            bits = (arrs["qual"][r0: r0 + ls].astype(np.int64) < P.min_gt_qual).astype(np.uint8)
            assert _same_low(q1, bits) and _same_low(q2, bits)
            if isinstance(q1, tuple):  # positions travel exactly with the records that can need them
                assert (q1[1] is not None) == (not no_seq[k] and q1[0] <= abi.QLOW_LIST_MAX)
            assert (s1 is None) == bool(no_seq[k]) and (s1 is None or np.array_equal(s1, s2))
        # the oracle on the selected records only
        sub = _subset_ascii(arrs, idx, nc)
        dvc = abi.dnms_view(dn.contig[a:b], dn.contig[a:b], dn.start[a:b], dn.end[a:b], np.zeros(b - a, np.uint8),
                            dn.refs[a:b], dn.alts[a:b], cutoff)
        fsub = orc.find(P, sh, fh, dvc, abi.FIND_SECOND_WINDOW)
        got = orc.phase(P, sh, sub, dvc, fsub, keep_lists=True)
        for k in ("status", "counts", "origin", "evidence"):
            assert np.array_equal(got[k], want[k][a:b]), (k, a, b)
        wo, wv, go, gv = want["vote_off"], want["vote_val"], got["vote_off"], got["vote_val"]
        for d in range(b - a):
            for j in range(4):
                assert np.array_equal(gv[go[4 * d + j]: go[4 * d + j + 1]], wv[wo[4 * (a + d) + j]: wo[4 * (a + d) + j + 1]])
    assert total < 0.6 * N  # about a third of the records inside the windows is reachable


def test_kernel_body_never_reads_the_bases_of_a_mate_only_record():
    """The rule behind UZ_AUX_NO_SEQ, checked on the kernel body itself (CPU twin, tests/emu): with every record that no
    fetch returns stripped of its bases, the extended read stage gives the oracle's results and never asks for a
    stripped record's bases; with --no-extended (no het-site fetches) the join DOES read mates at candidate sites, which
    is why such batches are staged with all_bases."""
    from emu import emu
    sc, dn, cl, rh, arrs = _workload(200, seed=17)
    n = dn.n
    sh, fh = _sites_views(sc)
    N = int(rh.view.n_segs)
    contig_of_rec = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
    cutoff = concordant_cutoff(arrs["tlen"], 151, 3)
    dv = abi.dnms_view(dn.contig, dn.contig, dn.start, dn.end, np.zeros(n, np.uint8), dn.refs, dn.alts, cutoff)
    tripped = {}
    for no_ext in (False, True):
        P = abi.make_params(no_extended=no_ext)
        found = orc.find(P, sh, fh, dv, abi.FIND_SECOND_WINDOW)
        want = orc.phase(P, sh, rh, dv, found, keep_lists=False)
        fc, flo, fhi = fetch_points(dn.contig, dn.start, np.zeros(n, np.uint8), sc.pos, found[3], found[4], P)
        direct = np.zeros(N, bool)
        for c in np.unique(fc):
            m = fc == c
            los = np.sort(flo[m])  # single-base fetches, plus [pos-1, pos+1) at the DNMs
            r0, r1 = arrs["contig_off"][c], arrs["contig_off"][c + 1]
            k = np.searchsorted(los, arrs["start"][r0:r1] - 1, side="left")
            direct[r0:r1] = (k < los.size) & (los[np.minimum(k, los.size - 1)] < arrs["end"][r0:r1])
        got = emu.phase(P, sh, rh, dv, found, no_seq=~direct)
        tripped[no_ext] = got["base_err"]
        if not no_ext:
            for k in ("status", "counts", "origin", "evidence"):
                assert np.array_equal(want[k], got[k]), k
            assert (want["status"] == abi.ST_OK).sum() > 20 and (~direct).mean() > 0.5
    assert tripped == {False: 0, True: 1}


def test_kernel_body_stays_inside_the_staged_units():
    """Unit masks (uz_reads_packed_view.umask): the selection keeps, of a record's rows, only the 32-base units that hold a fetched
    position (and the alleles of a DNM).  The kernel body (CPU twin) run on rows cut down by exactly those masks must give the
    oracle's results and never ask for a base of a unit that stayed home; with the masks of the het-site fetches alone (the DNM
    fetches left out) it must trip the guard instead."""
    from emu import emu
    sc, dn, cl, rh, arrs = _workload(220, seed=23)
    n = dn.n
    sh, fh = _sites_views(sc)
    N = int(rh.view.n_segs)
    P = abi.make_params()
    cutoff = concordant_cutoff(arrs["tlen"], 151, 3)
    dv = abi.dnms_view(dn.contig, dn.contig, dn.start, dn.end, np.zeros(n, np.uint8), dn.refs, dn.alts, cutoff)
    found = orc.find(P, sh, fh, dv, abi.FIND_SECOND_WINDOW)
    want = orc.phase(P, sh, rh, dv, found, keep_lists=False)
    assert (want["status"] == abi.ST_OK).sum() > 20
    alen = np.array([max(len(r), len(x)) for r, x in zip(dn.refs, dn.alts)], np.int64)
    assert alen.max() > 1  # indel DNMs: the alleles reach past the fetched position
    fc, flo, fhi, fex = fetch_points(dn.contig, dn.start, np.zeros(n, np.uint8), sc.pos, found[3], found[4], P, allele_len=alen)
    src = io_native.ReadsSource(io_native.pack_reads(rh, P.min_gt_qual, with_end=True))

    def masks_of(sel):
        part, idx = src.select(fc[sel], flo[sel], fhi[sel], want_index=True, extra=fex[sel], base_lists=False)  # (every staged unit as a row: the list form has its own test below)
        part.arrays.update(abi.small_columns(part)); part.arrays.update(abi.wide_columns(part))
        m = part.view.n_segs
        um = np.zeros(N, np.uint16)
        um[idx] = part.arrays["umask"][:m]
        no_seq = np.ones(N, bool)
        no_seq[idx] = (part.arrays["aux"][:m] & abi.AUX_NO_SEQ) != 0
        # the rows of the selection hold exactly the masked units of the source rows
        with_b = ~no_seq[idx]
        units = abi.row_units(part.arrays["l_seq"][:m])
        kept = np.where(part.arrays["umask"][:m] == abi.UMASK_ALL, units, [bin(int(x)).count("1") for x in part.arrays["umask"][:m]])
        assert part.view.n_seq_units == int(kept[with_b].sum()) < 0.5 * int(units[with_b].sum())
        # the listed low-quality positions: only those inside the staged units travel, and the count that travels with a listed
        # record is the length of that list (a record with more than ten keeps its full, saturated count and no list)
        full = src.packed
        true_low = full.arrays["n_low"][idx]
        n_low = part.arrays["n_low"][:m]
        listed = with_b & (true_low <= abi.QLOW_LIST_MAX)
        assert np.array_equal(n_low[~listed], true_low[~listed])
        off = np.concatenate([[0], np.cumsum(np.where(listed, n_low, 0).astype(np.int64))]).astype(np.int64)
        assert off[-1] == part.view.n_qlow_pos
        foff = np.concatenate([[0], np.cumsum(np.where(((full.arrays["aux"][:N] & abi.AUX_NO_SEQ) == 0) & (full.arrays["n_low"][:N] <= abi.QLOW_LIST_MAX),
                                                        full.arrays["n_low"][:N], 0).astype(np.int64))]).astype(np.int64)
        total_true = 0
        for k in np.nonzero(listed)[0][::7]:
            i = int(idx[k])
            all_pos = full.arrays["qlow_pos"][int(foff[i]): int(foff[i + 1])].astype(np.int64)
            mk = int(part.arrays["umask"][k])
            want_pos = all_pos if mk == abi.UMASK_ALL else all_pos[((mk >> (all_pos >> 5)) & 1) == 1]
            assert np.array_equal(part.arrays["qlow_pos"][int(off[k]): int(off[k + 1])].astype(np.int64), want_pos)
            total_true += all_pos.size
        assert off[-1] < 0.6 * int(true_low[listed].sum())
        return um, no_seq

    um, no_seq = masks_of(np.ones(fc.size, bool))
    got = emu.phase(P, sh, rh, dv, found, no_seq=no_seq, umask=um)
    assert got["base_err"] == 0
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(want[k], got[k]), k
    # masks that ignore the DNM fetches: the DNM-position units of most reads stay home -> the guard fires (err 3), loudly
    um2, _ = masks_of(np.arange(fc.size) >= n)  # (fetch_points lists the n DNM fetches first)
    got2 = emu.phase(P, sh, rh, dv, found, no_seq=no_seq, umask=np.where(no_seq, um, um2))
    assert got2["base_err"] == 3


def test_kernel_body_stays_inside_the_listed_bases():
    """The list form of the bases (uz_reads_packed_view.bl_*): a record that one- / two-base fetches return sends the bases at those
    positions (+ the DNM's alleles) instead of the 32-base units they lie in.  The kernel body (CPU twin) run on rows that hold ONLY the
    listed bases -- every other base of a staged unit reads as code 0, and a listed record refuses code 0 -- must give the oracle's results
    without tripping the guard; with the lists of the het-site fetches alone it must trip it (err 3 or 4).  And the form must pay:
    fewer link bytes than the units, codes equal to the source's bases."""
    from emu import emu
    sc, dn, cl, rh, arrs = _workload(220, seed=23)
    n = dn.n
    sh, fh = _sites_views(sc)
    N = int(rh.view.n_segs)
    P = abi.make_params()
    cutoff = concordant_cutoff(arrs["tlen"], 151, 3)
    dv = abi.dnms_view(dn.contig, dn.contig, dn.start, dn.end, np.zeros(n, np.uint8), dn.refs, dn.alts, cutoff)
    found = orc.find(P, sh, fh, dv, abi.FIND_SECOND_WINDOW)
    want = orc.phase(P, sh, rh, dv, found, keep_lists=False)
    alen = np.array([max(len(r), len(x)) for r, x in zip(dn.refs, dn.alts)], np.int64)
    fc, flo, fhi, fex = fetch_points(dn.contig, dn.start, np.zeros(n, np.uint8), sc.pos, found[3], found[4], P, allele_len=alen)
    src = io_native.ReadsSource(io_native.pack_reads(rh, P.min_gt_qual, with_end=True))

    def lists_of(sel):
        part, idx = src.select(fc[sel], flo[sel], fhi[sel], want_index=True, extra=fex[sel], base_lists=True)
        rows, _ = src.select(fc[sel], flo[sel], fhi[sel], want_index=True, extra=fex[sel], base_lists=False)
        small = abi.small_columns(part)
        m = int(part.view.n_segs)
        off, pos, code = abi.base_lists(part)
        cnt = np.diff(off)
        listed = cnt > 0
        assert listed.mean() > 0.3
        um_k = small["umask"][:m]
        pop = np.array([bin(int(x)).count("1") for x in um_k])
        # a listed record owns no row on the link; its units exist on the device only
        assert int(part.view.n_bl_units) == int(pop[listed].sum())
        assert int(part.view.n_seq_units) + int(part.view.n_bl_units) == int(rows.view.n_seq_units)
        assert 8 * int(part.view.n_seq_units) + 1.25 * int(part.view.n_bl) < 0.5 * 8 * int(rows.view.n_seq_units)  # it pays
        # every unit of a listed record's mask holds a listed base, every listed base lies in the mask, ascending
        rec_of = np.repeat(np.arange(m), cnt)
        assert (((um_k[rec_of].astype(np.int64) >> (pos.astype(np.int64) >> 5)) & 1) == 1).all()
        got_units = np.zeros(m, np.int64)
        np.bitwise_or.at(got_units, rec_of, 1 << (pos.astype(np.int64) >> 5))
        assert np.array_equal(got_units[listed], um_k[listed].astype(np.int64))
        same = rec_of[1:] == rec_of[:-1]
        assert (np.diff(pos.astype(np.int64))[same] > 0).all()
        # codes = the source's bases (A C G T -> 0 1 2 3)
        seq = arrs["seq"].reshape(-1, 160)
        src_base = seq[idx[rec_of], pos.astype(np.int64)]
        lut = np.full(256, 0, np.uint8); lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3
        assert np.array_equal(code, lut[src_base])
        um = np.zeros(N, np.uint16)
        um[idx] = um_k
        no_seq = np.ones(N, bool)
        no_seq[idx] = (small["aux"][:m] & abi.AUX_NO_SEQ) != 0
        full_off = np.zeros(N + 1, np.int64)
        full_cnt = np.zeros(N, np.int64)
        full_cnt[idx] = cnt
        full_off[1:] = np.cumsum(full_cnt)
        full_pos = np.zeros(int(full_off[-1]), np.uint16)
        for k in np.nonzero(listed)[0]:
            full_pos[full_off[idx[k]]: full_off[idx[k] + 1]] = pos[off[k]: off[k + 1]]
        return um, no_seq, (full_off, full_pos)

    um, no_seq, bl = lists_of(np.ones(fc.size, bool))
    got = emu.phase(P, sh, rh, dv, found, no_seq=no_seq, umask=um, bl=bl)
    assert got["base_err"] == 0
    for k in ("status", "counts", "origin", "evidence"):
        assert np.array_equal(want[k], got[k]), k
    um2, _, bl2 = lists_of(np.arange(fc.size) >= n)  # the DNM fetches left out: their bases are neither staged nor listed
    got2 = emu.phase(P, sh, rh, dv, found, no_seq=no_seq, umask=np.where(no_seq, um, um2), bl=bl2)
    assert got2["base_err"] in (3, 4)


def test_pair_form_codes_and_its_fallback():
    """tlen / mate / name id in one byte (uz_types.h, pair_d8): every code of the form on a table with odd records -- a pair whose
    template lengths are not its span, a record without a mate, a name carried by three records, mates further apart than a byte
    reaches -- decodes to the source's columns; a source whose name ids do not ascend by first appearance keeps the eight-bit
    differences (the pair form would renumber the names out of order)."""
    sc, dn, cl, rh, arrs = _workload(40)
    N = int(rh.view.n_segs)
    mate, tlen, qname = arrs["mate"], arrs["tlen"], arrs["qname"]
    firsts = np.nonzero(mate[:N] > np.arange(N))[0]
    a, b, c, d = (int(firsts[k]) for k in (3, 40, 90, 150))
    tlen[a] += 9; tlen[mate[a]] -= 9           # +-t for another t: the SECOND carries it
    tlen[b] += 5                               # not even symmetric: both records spelled out
    m = int(mate[c]); mate[c] = -1; mate[m] = -1  # no mates: a new name, then an old one
    qname[d + 1 if mate[d] != d + 1 else d + 2] = qname[d]  # a third record with the name (and its own pair broken by that)
    pk = io_native.pack_reads(rh, 20, with_end=True)
    src = io_native.ReadsSource(pk)
    contig_of = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
    fc = np.unique(contig_of).astype(np.int32)
    part, idx = src.select(fc, np.zeros(fc.size, np.int32), np.full(fc.size, 2 ** 31 - 1, np.int32), want_index=True)  # everything
    assert idx.size == N and "pair_d8" in part.arrays
    p = part.arrays["pair_d8"][:N]
    assert p[a] <= 252 and p[mate[a]] == 253 and {int(p[b]), int(p[mate[b]])} == {254, 255} and p[c] == 254 and p[m] == 255
    assert (p == 0).sum() > 0.9 * N / 2 and ((p >= 1) & (p <= 252)).sum() == ((p == 0) | (p == 253)).sum()
    w = abi.wide_columns(part)
    assert np.array_equal(w["start"], pk.arrays["start"][:N]) and np.array_equal(w["tlen"], tlen[:N]) and np.array_equal(w["mate"], mate[:N])
    assert np.array_equal(part.qname_map[w["qname"]], qname[:N])
    # far mates: keep one record in three of the source -- and shuffle nothing: the form still holds; then ids out of order
    far = np.arange(N)
    lo = pk.arrays["start"][:N][far[::97]].astype(np.int32)
    part2, idx2 = src.select(contig_of[::97].astype(np.int32), lo, lo + 1, want_index=True)
    w2 = abi.wide_columns(part2)
    assert np.array_equal(w2["tlen"], tlen[idx2]) and np.array_equal(part2.qname_map[w2["qname"]], qname[idx2])
    perm = np.random.default_rng(5).permutation(int(qname[:N].max()) + 1).astype(np.uint32)
    arrs["qname"][:N] = perm[qname[:N]]
    src3 = io_native.ReadsSource(io_native.pack_reads(rh, 20, with_end=True))
    part3, idx3 = src3.select(contig_of[::97].astype(np.int32), lo, lo + 1, want_index=True)
    assert "pair_d8" not in part3.arrays and "mate_d8" in part3.arrays and part3.qname_map is None
    w3 = abi.wide_columns(part3)
    assert np.array_equal(idx3, idx2) and np.array_equal(w3["qname"], arrs["qname"][idx3]) and np.array_equal(w3["tlen"], tlen[idx3])


def test_span_sums_of_a_packed_view_are_the_running_sums_of_its_columns():
    """uz_packed_block_sums (uz_types.h pk_sums) against numpy on a link-form selection: per span of 1024 records the running sums of the eleven
    quantities the device's header build lays the records out by (csrc/k_reads.hip pk_vals)."""
    from synth.small import SmallConfig, make_small
    from helpers import tables
    ds = make_small(SmallConfig(seed=4242, n_dnms=40, cluster_prob=0.7, indel_prob=0.05, softclip_prob=0.05))
    _, reads = tables(ds)
    kid = sorted(reads)[0]
    full = io_native.pack_reads(abi.reads_view(reads[kid]), 20, lists=True, with_end=True)
    st = np.asarray(reads[kid].start)
    pts = np.sort(np.random.default_rng(3).choice(st, 400))
    fc = np.asarray(reads[kid].contig_of(pts) if hasattr(reads[kid], "contig_of") else np.zeros(pts.size), np.int32)
    sel = io_native.ReadsSource(full).select(fc, pts.astype(np.int32), (pts + 1).astype(np.int32), extra=np.zeros(pts.size, np.uint16))
    v, a = sel.view, sel.arrays
    n = int(v.n_segs)
    assert n > 1500 and "pk_sums" in a and int(v.n_pk_spans) == abi.pk_spans(n) == (n + 1023) // 1024
    S = a["pk_sums"].reshape(-1, abi.PK_SUMS)
    t = abi.tup_column(sel).astype(np.int64)
    ls, nc, ax = a["tup_l_seq"][t].astype(np.int64), a["tup_n_cigar"][t].astype(np.int64), a["tup_aux"][t].astype(np.int64)
    nl, um = a["tup_n_low"][t].astype(np.int64), a["tup_umask"][t].astype(np.int64)
    nb = a["tup_n_bl"][t].astype(np.int64) if "tup_n_bl" in a else np.zeros(n, np.int64)
    units = (ls + 31) >> 5
    staged = np.where(ax & abi.AUX_NO_SEQ, 0, np.where(um == abi.UMASK_ALL, units, np.array([bin(int(x)).count("1") for x in um])))
    pd = a["pair_d8"][:n].astype(np.int64)
    sd = a["start_d8"][:n].astype(np.int64)
    key, val = a["esc16_key"][: int(v.n_esc16)], a["esc16_val"][: int(v.n_esc16)]
    esc0 = {int(k >> 2): int(x) for k, x in zip(key, val) if int(k) & 3 == 0}
    sdv = np.array([esc0.get(i, 0) if sd[i] == 255 else sd[i] for i in range(n)], np.int64) & 0xFFFFFFFF
    q = np.zeros((n, abi.PK_SUMS), np.int64)
    q[:, 0], q[:, 1] = nc, units
    q[:, 2], q[:, 7], q[:, 8] = np.where(nb > 0, 0, staged), np.where(nb > 0, staged, 0), nb
    q[:, 3] = np.where((ax & abi.AUX_NO_SEQ) == 0, np.where(nl <= abi.QLOW_LIST_MAX, nl, 0), 0)
    q[:, 4] = np.where(ax & abi.AUX_SIMPLE_MASK, 0, nc)
    q[:, 5] = sdv
    q[:, 6] = ((pd != 0) & (pd != 253) & (pd != 255)).astype(np.int64)
    q[:, 9], q[:, 10] = ((pd >= 1) & (pd <= 252)).astype(np.int64), ((pd == 0) | (pd == 253)).astype(np.int64)
    run = np.concatenate([np.zeros((1, abi.PK_SUMS), np.int64), np.cumsum(q, axis=0)])
    want = run[np.minimum(np.arange(S.shape[0]) * 1024, n)]
    assert np.array_equal(S.astype(np.int64), want)
    assert S[-1, 0] == int(v.n_cigar_total) + int(v.n_cigar_omitted) and S[-1, 1] == int(v.n_row_units) and S[-1, 9] == S[-1, 10]


def test_the_dictionary_index_in_one_byte_round_trips():
    """abi.compact_tup (uz_types.h tup8): the 255 most frequent combinations by a byte, the rest through the escape list, escapes counted per
    span of 1 024 records -- rebuilt, the 16-bit index is the one that went in; a table with more than 255 combinations in use escapes some"""
    rng = np.random.default_rng(11)
    for n, n_tup, skew in ((0, 1, 1.0), (1, 1, 1.0), (1023, 40, 1.0), (1024, 300, 1.2), (1025, 3000, 1.1), (70001, 5000, 1.05)):
        p = 1.0 / np.arange(1, n_tup + 1) ** skew
        tup = rng.choice(n_tup, size=n, p=p / p.sum()).astype(np.uint16)
        v = abi.ReadsPackedView()
        arr = np.zeros(max(1, n), np.uint16)
        arr[:n] = tup
        v.n_segs, v.n_tup, v.tup = n, n_tup, arr.ctypes.data
        held = abi.Held(v, {"tup": arr})
        assert abi.compact_tup(held)
        a = held.arrays
        assert "tup" not in a and not held.view.tup and held.view.tup8
        assert np.array_equal(abi.tup_column(held), tup)
        n_esc = int(held.view.n_tup_esc)
        assert (n_esc > 0) == (np.unique(tup).size > 255)
        nsp = (n + abi.TUP8_SPAN - 1) // abi.TUP8_SPAN
        off = a["tup_esc_off"][: nsp + 1]
        assert off[0] == 0 and int(off[-1]) == n_esc and np.all(np.diff(off.astype(np.int64)) >= 0)
        for b in range(nsp):
            assert int(off[b + 1] - off[b]) == int((a["tup8"][b * abi.TUP8_SPAN: min(n, (b + 1) * abi.TUP8_SPAN)] == 255).sum())
