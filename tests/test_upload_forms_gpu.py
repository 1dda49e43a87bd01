"""What the device makes of a table's fixed-width columns (uz_reads_headers) equals the source's columns in every form they can
travel in: plain, 16-bit differences, 8-bit start / mate / name-id differences, the pair form (one byte for tlen, mate and name id)
with every one of its codes -- FIRST / SECOND, a SECOND that brings its own template length, records spelled out in the escape list,
with a new or an old name -- and straight from an indexed BAM (uz_bam_stage_*).  A pair form that contradicts itself is refused."""
import numpy as np
import pytest

from test_pack_select import _workload
from unfazed_amd import abi, io_native
from unfazed_amd.engine import UnfazedHipError

pytestmark = pytest.mark.gpu


def _odd_table(n_dnms=40):
    sc, dn, cl, rh, arrs = _workload(n_dnms)
    N = int(rh.view.n_segs)
    mate, tlen, qname = arrs["mate"], arrs["tlen"], arrs["qname"]
    firsts = np.nonzero(mate[:N] > np.arange(N))[0]
    a, b, c, d = (int(firsts[k]) for k in (3, 40, 90, 150))
    tlen[a] += 9; tlen[mate[a]] -= 9
    tlen[b] += 5
    m = int(mate[c]); mate[c] = -1; mate[m] = -1
    qname[d + 1 if mate[d] != d + 1 else d + 2] = qname[d]
    return rh, arrs, N


def test_every_form_decodes_to_the_source_columns(engine):
    rh, arrs, N = _odd_table()
    pk = io_native.pack_reads(rh, 20, with_end=True)
    src = io_native.ReadsSource(pk)
    contig_of = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
    fc = np.unique(contig_of).astype(np.int32)
    everything = (fc, np.zeros(fc.size, np.int32), np.full(fc.size, 2 ** 31 - 1, np.int32))
    lo = arrs["start"][:N][::53].astype(np.int32)
    some = (contig_of[::53].astype(np.int32), lo, lo + 1)
    seen = set()
    for fetches in (everything, some):
        for kw in (dict(d16=False), dict(start8=False), dict(pair8=False, narrow8=False), dict(pair8=False), dict()):
            part, idx = src.select(*fetches, want_index=True, **kw)
            seen.add(tuple(sorted(k for k in part.arrays if k.endswith(("_d", "_d8", "_s")))))
            rid = engine.upload_reads_packed(part)
            got = engine.reads_headers(rid, idx.size)
            engine.free_reads(rid)
            new_of = np.full(N, -1, np.int64)
            new_of[idx] = np.arange(idx.size)
            m = arrs["mate"][:N][idx]
            want_mate = np.where(m >= 0, new_of[np.maximum(m, 0)], -1)
            assert np.array_equal(got["start"], arrs["start"][:N][idx]) and np.array_equal(got["end"], arrs["end"][:N][idx]), kw
            assert np.array_equal(got["tlen"], arrs["tlen"][:N][idx]), kw
            assert np.array_equal(got["mate"], want_mate), kw
            q = got["qname"] if part.qname_map is None else part.qname_map[got["qname"]]
            assert np.array_equal(q, arrs["qname"][:N][idx]), kw
            if "pair_d8" in part.arrays:
                p = part.arrays["pair_d8"][: idx.size]
                assert set(np.unique(p[(p == 0) | (p > 252)]).tolist()) >= ({0, 253, 254, 255} if idx.size == N else {0})
    assert len(seen) == 5


def test_a_pair_form_that_contradicts_itself_is_refused(engine):
    rh, arrs, N = _odd_table(20)
    src = io_native.ReadsSource(io_native.pack_reads(rh, 20, with_end=True))
    contig_of = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
    fc = np.unique(contig_of).astype(np.int32)
    for breakage in ("second_named_twice", "orphan_second", "first_names_a_first"):
        part = src.select(fc, np.zeros(fc.size, np.int32), np.full(fc.size, 2 ** 31 - 1, np.int32))
        p = part.arrays["pair_d8"]
        f = np.nonzero((p[:N] >= 1) & (p[:N] <= 252))[0]
        if breakage == "second_named_twice":  # two FIRST records name one SECOND (and another SECOND is left over)
            i, j = int(f[5]), int(f[5]) + int(p[f[5]])
            k = next(int(x) for x in f if x != i and 0 < j - int(x) <= 252)
            p[k] = j - k
        elif breakage == "orphan_second":     # a FIRST turned into a spelled-out record would need escapes: turn it into a SECOND instead
            p[int(f[7])] = 0
        else:
            i = int(f[9])
            k = next(int(x) for x in f if 0 < int(x) - i <= 252 and int(x) != i + int(p[i]))
            p[i] = k - i
        with pytest.raises(UnfazedHipError):
            rid = engine.upload_reads_packed(part)
            engine.wait_reads(rid)


def test_a_base_list_that_contradicts_its_mask_is_refused(engine):
    """The list form of the bases (uz_types.h bl_*): the header build lays a listed record's units out from its mask and writes the listed
    bases into them -- a position outside the mask, positions that do not ascend, or a unit of the mask nobody lists is a table whose
    columns disagree, refused when the table is first used (never a base written somewhere else)."""
    sc, dn, cl, rh, arrs = _workload(30, seed=5)
    N = int(rh.view.n_segs)
    src = io_native.ReadsSource(io_native.pack_reads(rh, 20, with_end=True))
    # one-base fetches at every 97th record's 40th base: single positions -> lists
    idx = np.arange(0, N, 97)
    contig_of = np.searchsorted(arrs["contig_off"], idx, side="right") - 1
    lo = (arrs["start"][:N][idx] + 40).astype(np.int32)
    fetches = (contig_of.astype(np.int32), lo, lo + 1)
    ex = np.ones(idx.size, np.uint16)  # (+ one base on: two listed bases per record)

    def fresh():
        part = src.select(*fetches, extra=ex, base_lists=True)
        off, pos, code = abi.base_lists(part)
        assert int(part.view.n_bl) > 50 and not part.view.bl_wide
        return part, off
    part, off = fresh()
    rid = engine.upload_reads_packed(part)  # the honest table is taken
    engine.wait_reads(rid)
    engine.free_reads(rid)
    k = int(np.nonzero(np.diff(off) >= 1)[0][3])
    for breakage in ("outside_the_mask", "not_ascending", "beyond_the_read"):
        part, off = fresh()
        bp = part.arrays["bl_pos"]
        if breakage == "outside_the_mask":
            bp[off[k]] = (int(bp[off[k]]) + 64) % 150  # two units further: not in the record's mask
        elif breakage == "not_ascending":
            kk = int(np.nonzero(np.diff(off) >= 2)[0][0])
            bp[off[kk] + 1] = bp[off[kk]]
        else:
            bp[off[k]] = 200  # l_seq is 151
        with pytest.raises(UnfazedHipError):
            rid = engine.upload_reads_packed(part)
            engine.wait_reads(rid)


def test_staged_from_an_indexed_bam(engine, tmp_path):
    from synth import bigsynth
    from synth.sites_np import make_clusters, make_sites, place_dnms_full
    sc = make_sites(40_000, seed=7, contig_lens=[6e6, 4e6, 2e6])
    dn = place_dnms_full(sc, 60, seed=8, indel_frac=0.2)
    cl = make_clusters(dn)
    cfg = bigsynth.make_cfg(seed=9)
    cfg.n_clusters = cl.n
    bam = str(tmp_path / "kid.bam")
    bigsynth.write_bam(bam, cfg, sc, dn, cl, 0, cl.n, level=1)
    table = io_native.read_bam_table(bam, threads=2)
    srcb = io_native.BamSource(bam, threads=2)
    tid = dn.contig[::2].astype(np.int32)
    lo = (dn.start[::2] - 1).astype(np.int32)
    got = srcb.select(tid, lo, lo + 2, 20, extra=np.zeros(tid.size, np.uint16))
    n = int(got.view.n_segs)
    assert "pair_d8" in got.arrays and n > 1000
    rid = engine.upload_reads_packed(got)
    dev = engine.reads_headers(rid, n)
    engine.free_reads(rid)
    w = abi.wide_columns(got)
    for k in ("start", "tlen", "mate", "qname"):
        assert np.array_equal(dev[k], w[k]), k
    assert np.array_equal(dev["end"], abi.record_ends(got, w["start"].astype(np.int64)))
    # ... and those are the file's records: names and template lengths by name
    by_name = {}
    for i in range(table.start.size):
        by_name.setdefault(table.qnames[int(table.qname[i])], []).append((int(table.start[i]), int(table.tlen[i])))
    for i in range(0, n, 37):
        assert (int(dev["start"][i]), int(dev["tlen"][i])) in by_name[got.qnames[int(dev["qname"][i])]]


def test_family_columns_in_eight_bits_give_the_same_classes(engine):
    """The nine genotype columns of a trio in the eight-bit link form (uz_family_view.ref_depth8 ...: depths below 254, 254 = missing,
    255 = the site stands in the wide list; qualities clamped at 254, 255 = missing) against the 16-bit columns: the same class byte at
    every site, on both upload routes and for several threshold sets; a --min-gt-qual the clamp cannot serve is refused."""
    from synth.sites_np import make_sites
    from test_site_stage_gpu import PARAM_SETS, _Sites
    from test_pack_select import _sites_views  # noqa: F401  (shared helpers live there)
    sc = make_sites(60_001, seed=91)
    rd, ad, gq = (np.array(x, np.uint16, copy=True) for x in (sc.rd, sc.ad, sc.gq))
    rng = np.random.default_rng(3)
    deep = rng.choice(sc.n, 300, replace=False)
    rd[0, deep[:100]] = rng.integers(254, 2000, 100)     # depths the bytes cannot hold: through the wide list
    ad[1, deep[100:200]] = rng.integers(254, 30000, 100)
    ad[2, deep[200:]] = 254
    gq[1, deep[:50]] = rng.integers(255, 1000, 50)       # qualities beyond the clamp
    assert (rd == 0xFFFF).any() and (gq == 0xFFFF).any()  # missing values are part of the table
    r8, a8, g8, wide = abi.family_columns8(rd, ad, gq)
    assert r8.dtype == np.uint8 and len(wide[0]) >= 290 and (r8 == 254).sum() == (rd == 0xFFFF).sum() - int((rd[:, wide[0]] == 0xFFFF).sum())
    sid = engine.upload_sites(_Sites(sc))
    f16 = engine.add_family(sid, sc.gt, rd, ad, gq)
    f8 = engine.add_family(sid, sc.gt, r8, a8, g8, wide)
    sv = abi.SitesView()
    keep = dict(contig_off=np.ascontiguousarray(sc.contig_off, np.int64), pos=sc.pos, sflags=sc.sflags, ref_base=sc.ref_base, alt_base=sc.alt_base)
    sv.n_sites, sv.n_contigs = sc.n, len(sc.contig_off) - 1
    for k, a in keep.items():
        setattr(sv, k, a.ctypes.data)
    sid2, f8a = engine.upload_sites_family_async(abi.Held(sv, keep), sc.gt, list(r8), list(a8), list(g8), wide)
    for kw in PARAM_SETS:
        P = abi.make_params(**kw)
        want = engine.classify(f16, P, sc.n)
        assert np.array_equal(engine.classify(f8, P, sc.n), want), kw
        assert np.array_equal(engine.classify(f8a, P, sc.n), want), kw
    assert len(np.unique(want)) > 3
    with pytest.raises(UnfazedHipError, match="eight-bit genotype qualities"):
        engine.classify(f8, abi.make_params(min_gt_qual=300), sc.n)
    engine.classify(f16, abi.make_params(min_gt_qual=300), sc.n)  # (the 16-bit columns serve any threshold)
    engine.free_sites(sid)
    engine.free_sites(sid2)


def test_span_sums_from_the_packer(engine):
    """uz_types.h pk_sums: the header build packs from the packer's span sums (no k_off_block_sums / scan) and must build the same table as from its
    own; sums that do not add up -- one span's share moved to its neighbour, or totals that differ from the view's -- are refused."""
    rh, arrs, N = _odd_table(60)
    pk = io_native.pack_reads(rh, 20, with_end=True)
    src = io_native.ReadsSource(pk)
    contig_of = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
    fc = np.unique(contig_of).astype(np.int32)
    part, idx = src.select(fc, np.zeros(fc.size, np.int32), np.full(fc.size, 2 ** 31 - 1, np.int32), want_index=True)
    n = idx.size
    assert "pk_sums" in part.arrays and int(part.view.n_pk_spans) >= 3
    S = part.arrays["pk_sums"].reshape(-1, abi.PK_SUMS)
    good = S.copy()
    rid = engine.upload_reads_packed(part)
    with_sums = engine.reads_headers(rid, n)
    engine.free_reads(rid)
    keep = part.view.pk_sums
    part.view.pk_sums = None  # the device counts for itself
    rid = engine.upload_reads_packed(part)
    own = engine.reads_headers(rid, n)
    engine.free_reads(rid)
    for k in ("start", "end", "tlen", "mate", "qname"):
        assert np.array_equal(with_sums[k], own[k]), k
    part.view.pk_sums = keep
    # a CIGAR word moved from span 1 to span 2: the rows still ascend and end at the totals, but span 1 does not add up to its row
    S[2, 0] -= 1
    with pytest.raises(UnfazedHipError):
        rid = engine.upload_reads_packed(part)
        engine.wait_reads(rid)
        engine.reads_headers(rid, n)
    S[:] = good
    S[2, 2] += 1  # ... one base-row unit the other way
    with pytest.raises(UnfazedHipError):
        rid = engine.upload_reads_packed(part)
        engine.wait_reads(rid)
        engine.reads_headers(rid, n)
    S[:] = good
    S[-1, 1] += 1  # totals that are not the view's: refused before anything is copied
    with pytest.raises(UnfazedHipError, match="pk_sums"):
        engine.upload_reads_packed(part)
    S[:] = good
    rid = engine.upload_reads_packed(part)  # ... and the untouched sums still go through
    again = engine.reads_headers(rid, n)
    engine.free_reads(rid)
    assert np.array_equal(again["start"], own["start"])


def test_the_dictionary_index_in_one_byte(engine):
    """uz_types.h tup8: the device rebuilds the 16-bit index from the byte column, the hot table and the escape list before the header build reads
    it -- the same table as from `tup` (headers; the read stage's answers are held against the resident pass by every staged test of the suite) --
    and refuses a form that contradicts itself: escapes that are not where the span offsets say, an index beyond the dictionary."""
    rh, arrs, N = _odd_table(60)
    pk = io_native.pack_reads(rh, 20, with_end=True)
    src = io_native.ReadsSource(pk)
    contig_of = np.searchsorted(arrs["contig_off"], np.arange(N), side="right") - 1
    fc = np.unique(contig_of).astype(np.int32)
    fetches = (fc, np.zeros(fc.size, np.int32), np.full(fc.size, 2 ** 31 - 1, np.int32))
    plain, idx = src.select(*fetches, want_index=True, tup8=False)
    n = idx.size
    assert "tup" in plain.arrays
    rid = engine.upload_reads_packed(plain)
    want = engine.reads_headers(rid, n)
    engine.free_reads(rid)

    def fresh(hot_keep=None):
        part, _ = src.select(*fetches, want_index=True, tup8=False)
        if hot_keep is not None:  # (few enough hot entries that the escape list is exercised on this small table)
            t = part.arrays["tup"][:n]
            keep = np.argsort(-np.bincount(t, minlength=int(part.view.n_tup)), kind="stable")[:hot_keep]
            part.arrays["tup"][:n] = t  # unchanged; the cut happens below
            assert abi.compact_tup(part)
            a = part.arrays
            full = abi.tup_column(part)
            # rebuild the form with only `hot_keep` hot entries
            lut = np.full(int(part.view.n_tup), 255, np.uint8)
            lut[keep] = np.arange(keep.size, dtype=np.uint8)
            a["tup8"][:n] = lut[full]
            esc = a["tup8"][:n] == 255
            ne = int(esc.sum())
            e16 = np.zeros(max(1, ne), np.uint16)
            e16[:ne] = full[esc]
            off = np.zeros((n + abi.TUP8_SPAN - 1) // abi.TUP8_SPAN + 1, np.uint32)
            off[1:] = np.cumsum(np.add.reduceat(esc.astype(np.int64), np.arange(0, n, abi.TUP8_SPAN)))
            a["tup_hot"][:] = 0
            a["tup_hot"][: keep.size] = keep
            a["tup_esc"], a["tup_esc_off"] = e16, off
            part.view.tup_esc, part.view.tup_esc_off, part.view.n_tup_esc = e16.ctypes.data, off.ctypes.data, ne
        else:
            assert abi.compact_tup(part)
        return part
    for hot_keep in (None, 3):
        part = fresh(hot_keep)
        assert "tup8" in part.arrays and "tup" not in part.arrays and (hot_keep is None or int(part.view.n_tup_esc) > 100)
        rid = engine.upload_reads_packed(part)
        got = engine.reads_headers(rid, n)
        engine.free_reads(rid)
        for k in ("start", "end", "tlen", "mate", "qname"):
            assert np.array_equal(got[k], want[k]), (k, hot_keep)
    # an escape moved from one span to the next: the offsets still ascend to the total, but the spans do not hold what they say
    part = fresh(3)
    off = part.arrays["tup_esc_off"]
    assert off.size >= 3 and off[1] > 0
    off[1] -= 1
    with pytest.raises(UnfazedHipError):
        rid = engine.upload_reads_packed(part)
        engine.wait_reads(rid)
        engine.reads_headers(rid, n)
    part = fresh(3)
    part.arrays["tup_esc"][0] = 65535  # an index beyond the dictionary
    with pytest.raises(UnfazedHipError):
        rid = engine.upload_reads_packed(part)
        engine.wait_reads(rid)
        engine.reads_headers(rid, n)
    part = fresh(3)
    part.arrays["tup_esc_off"][-1] += 1  # offsets that do not end at the total: refused before anything is copied
    with pytest.raises(UnfazedHipError):
        engine.upload_reads_packed(part)
    part = fresh(3)  # ... and the untouched form still goes through
    rid = engine.upload_reads_packed(part)
    engine.wait_reads(rid)
    engine.free_reads(rid)
