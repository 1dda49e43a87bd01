// phase_body.hpp -- the per-DNM read stage as block-parallel phases (see wg.hpp).
//
// One workgroup phases one DNM at a time:
//   A  DNM reads         collect_reads_snv :382-425 (+ snv/indel_match_alleles :266-336)
//   B  registration      group_reads_by_haplotype :165-222 (fetch at every het site)
//   C  seeding           :223-249
//   S  pair table        one sort of (qname, sequence) keys gives dense pair ids, the
//                        per-pair read_sites lists in append order and "last writer wins"
//                        for fetched_reads (quirk Q11)
//   D  static tables     finder / candidate alleles per list entry (get_allele_at :56-73)
//   E  chaining          connect_reads :76-152 as a level-synchronous BFS; every unassigned
//                        pair takes the attempt of minimum sequential rank (64-bit atomicMin),
//                        winners sorted by rank form the next level
//   F  join + vote       match_informative_sites (site_searcher.py:50-78), phase_by_reads
//                        (snv_phaser.py:16-70), unique-name / unique-site tally (:169-185) and
//                        the integer decision of summarize_record (unfazed.py:206-234)
// (file names relative to reference unfazed/; read_collector.py unless stated)
#pragma once
#include "uz_types.h"
#include "pack.hpp"
#include "wg.hpp"

// UZ_TICK(k) marks the boundary between two phases of the body: the arguments and scratch pointers of the next phase are read afresh there
// (UZ_PHASE_ARGS, below).  Diagnostic build only (-DUZ_PHASE_TIMING): lane 0 also adds the shader-clock ticks spent since the last
// boundary into a.timing[k]; never compiled into the product library.
#if defined(UZ_PHASE_TIMING) && !defined(UZ_EMU)
#define UZ_TICK(k)                                                                     \
    do {                                                                               \
        __syncthreads();                                                               \
        UZ_PHASE_ARGS();                                                               \
        if (threadIdx.x == 0) { /* summed per wave in LDS, flushed once when the wave exits (k_phase): 17 atomics per DNM on 17 addresses bound the whole kernel */ \
            const unsigned long long now__ = __builtin_amdgcn_s_memtime();             \
            sh->tick[k] += now__ - tick__;                                             \
            tick__ = now__;                                                            \
        }                                                                              \
    } while (0)
#define UZ_TICK_INIT unsigned long long tick__ = __builtin_amdgcn_s_memtime()
#else
#define UZ_TICK(k) UZ_PHASE_ARGS()
#define UZ_TICK_INIT ((void)0)
#endif

#ifndef UZ_BW_SITES
#define UZ_BW_SITES 1 // items (het sites, a point variant's own fetch) a lane of the sizing pass searches side by side (uz_phase_bounds_w)
#endif
#ifndef UZ_PHASE_K
#define UZ_PHASE_K 3 // entries a lane works on at once in the finder pass (phase D): their loads are in flight together (2 / 3 / 4: 3.02 / 2.97 / 3.09 ms per 100 k DNMs; 4 spills vector registers)
#endif
#define UZ_QC_GOOD 1u      // goodread(read) :28-53
#define UZ_QC_GOOD_DISC 2u // goodread(read, True)
#define UZ_QC_NM5 4u       // <= 5 CIGAR ops other than M/= :190-196
#define UZ_QC_NONE5 8u     // <= 5 query bases without a reference position :200-203, :405-408

#define UZ_OP_M 0
#define UZ_OP_I 1
#define UZ_OP_D 2
#define UZ_OP_N 3
#define UZ_OP_S 4
#define UZ_OP_EQ 7
#define UZ_OP_X 8

#ifdef UZ_EMU_STATS
extern "C" long long uz_emu_stats[16];
extern "C" long long *uz_emu_log; // optional: 12 values per DNM (sizes of its working set)
#endif
// Record headers: the form the fixed-width fields of the staged columns (uz_reads_packed_view) take in HBM,
// built on the device when a table is uploaded / adopted.  The per-DNM kernel GATHERS records (mates, last
// registrations, init elements), and a gather that finds start / end / offsets / lengths in one or two
// 16-byte words touches one or two cache lines instead of five to eight.  cigar_off / sq_off are the prefix
// sums of n_cigar / UZ_ROW_UNITS(l_seq) (rows and CIGAR words lie back to back in record order).
struct RecA { int32_t start, end; uint32_t cigar_off, sq_off; }; // sq_off: seq4 row, in row units (32 bases); UZ_NO_SEQ_OFF = staged without bases
#define UZ_NO_SEQ_OFF 0xFFFFFFFFu
#define UZ_NO_QLOW_OFF 0xFFFFFFFFu
struct RecB { int32_t mate; uint32_t qname; uint16_t l_seq, n_cigar; int32_t tlen; };
UZ_HD void uz_pack_rec(RecA &A, RecB &B, int32_t start, int32_t end, uint32_t cigar_off, uint32_t sq_off, int32_t mate,
                       uint32_t qname, uint16_t l_seq, uint16_t n_cigar, int32_t tlen) {
    A.start = start; A.end = end; A.cigar_off = cigar_off; A.sq_off = sq_off;
    B.mate = mate; B.qname = qname; B.l_seq = l_seq; B.n_cigar = n_cigar; B.tlen = tlen;
}
// flag | mapq << 16 | aux << 24
UZ_HD uint32_t uz_pack_fm(uint32_t flag, uint32_t mapq, uint32_t aux) { return (flag & 0xFFFFu) | ((mapq & 0xFFu) << 16) | ((aux & 0xFFu) << 24); }

struct RD { // alignment records of one table (device pointers)
    const RecA *ra;
    const RecB *rb;
    const uint32_t *fm; // flag | mapq << 16 | aux << 24
    const int64_t *contig_off;
    const int32_t *max_span;
    int32_t n_contigs;
    const uint32_t *cigar; // BAM encoding, back to back in record order
    const uint8_t *seq4;   // 4-bit bases, 16 bytes per row unit
    const uint8_t *qlow;   // 1 bit per base (quality below the threshold), 4 bytes per row unit
    const uint32_t *qoff;  // quality-plane row of every record, in row units; UZ_NO_QLOW_OFF = no row (the list form of the staged plane keeps
                           // rows only for the records whose bits can be asked for: bases staged and at most UZ_QLOW_LIST_MAX low ones)
    const uint8_t *nlow;   // number of low-quality bases of every record, saturated at 255
    const uint16_t *umask; // which 32-base units of a record's rows were staged: bit u = unit u, UZ_UMASK_ALL = every unit (rows hold the
                           // staged units back to back); a base or a quality bit of a unit that stayed home sets err = 3
    int32_t *err;          // [0] set when the bases of a record staged without them are requested (must never happen)
    const uint16_t *qs;    // per-record QC word, built with the headers: bits 0-3 the parameter-independent part of the QC bits (UZ_QC_*:
                           // GOOD / GOOD_DISC as they are when the mapping quality passes), bits 8-15 the mapping quality (uz_qc applies
                           // --min-map-qual where the bits are used: no per-batch QC pass)
    int32_t min_map_qual;
    const int32_t *coarse; // start of every 4096th record (L2-resident search index), may be null
    const int32_t *mid;    // start of every 64th record (second level of the search index: 4 B per 64 records, 64 entries = two lines per coarse cell), may be null
    const int32_t *mid8;   // start of every 8th record (third level: the eight entries of a 64-record cell are 32 bytes, the eight headers under one of them ONE line), may be null
};

// ---- per-record QC (goodread read_collector.py:28-53 and the two CIGAR counts of :190-203, :405-408)
UZ_HD void uz_cigar_op_counts(uint32_t c, int &nonmatch, int &none) {
    const int op = c & 15, l = (int)(c >> 4);
    if (op != UZ_OP_M && op != UZ_OP_EQ) nonmatch++;
    if (op == UZ_OP_I || op == UZ_OP_S) none += l;
}
// the QC word of a record: everything but the comparison with --min-map-qual, which the readers of the bits apply
// (low: bases below the base-quality threshold; nc: CIGAR operations; nonmatch / none: the two counts above)
UZ_HD uint16_t uz_qs_word(uint32_t f, uint32_t aux, uint32_t mapq, int low, int nc, int nonmatch, int none) {
    if (aux & UZ_AUX_DECODE_BAD) return 0; // no CIGAR / SEQ / QUAL: never a good read (unpinned)
    const bool flags_ok = !((f & 512u) || (f & 4u) || (f & 1024u) || (f & 256u) || (f & 2048u) || (f & 8u) || !(aux & UZ_AUX_MATE_SAME_TID)); // :31-41
    uint32_t q = 0;
    if (flags_ok) {
        q |= UZ_QC_GOOD_DISC;
        if (low <= 10 && nc <= 10) q |= UZ_QC_GOOD; // :43-52; "mismatches" counts every CIGAR op (quirk Q9)
    }
    if (nonmatch <= 5) q |= UZ_QC_NM5;
    if (none <= 5) q |= UZ_QC_NONE5;
    return (uint16_t)(q | ((mapq & 0xFFu) << 8));
}
// QC bits of a record for the run's --min-map-qual
UZ_HD uint32_t uz_qc_of(uint32_t w, int min_map_qual) {
    return ((int)(w >> 8) < min_map_qual) ? (w & (UZ_QC_NM5 | UZ_QC_NONE5)) : (w & 15u);
}

struct Caps { // per-workgroup scratch capacities (elements)
    int32_t A, T, H, C, I, M;
};

// Element types of the working arrays that differ between the two builds of the per-DNM body (see Arena below): the
// HBM build must take any DNM the capacities admit (32-bit indices, 64-bit sort and rank keys); a DNM that fits an LDS arena has at
// most 127 het sites, fewer than 64 k pair-table entries, 8 k registrations and 1 k init elements, and its arrays take the narrowest
// type that holds them -- the arena of a DNM is what decides how many DNMs a CU works on at once.
template <bool LDS> struct ScrTy {
    typedef int32_t hidx; typedef int32_t pidx; typedef int32_t xidx; typedef int32_t rlen; typedef uint32_t gflg; typedef int32_t fidx;
    typedef unsigned long long skey; typedef unsigned long long rkey;
};
template <> struct ScrTy<true> {
    typedef int8_t hidx; typedef uint16_t pidx; typedef uint16_t xidx; typedef uint8_t rlen; typedef uint8_t gflg; typedef uint16_t fidx;
    typedef uint32_t skey; typedef uint32_t rkey;
};
// A pair-table entry after the sort is ONE word (the array of the sort keys, rewritten in place): its sequence number, its het index, the
// finder allele the pair shows at that site as a code (0 none, 1 the site's REF, 2 its ALT: get_allele_at :91-105) and the base the pair's
// primary segment shows there (ASCII, 0 = none: :114-124).  Sequence numbers keep the reference's time order without being dense:
// registration k has k, seed j has T + j, the presence entry of init element m has T + S + m (T = records the het-site fetches returned).
template <bool LDS> struct En {
    typedef unsigned long long W;
    static constexpr int SEQ_B = 24, H_B = 20; // 24 + 20 + 2 + 8 bits of 64
};
template <> struct En<true> {
    typedef uint32_t W;
    static constexpr int SEQ_B = 15, H_B = 7; // 15 + 7 + 2 + 8 = 32 (the DNM is given up when T + S + nI or nh would not fit)
};
template <bool LDS> UZ_DEV typename En<LDS>::W uz_en_make(int seq, int h) { return (typename En<LDS>::W)seq | ((typename En<LDS>::W)h << En<LDS>::SEQ_B); }
template <bool LDS> UZ_DEV int uz_en_seq(typename En<LDS>::W w) { return (int)(w & (((typename En<LDS>::W)1 << En<LDS>::SEQ_B) - 1)); }
template <bool LDS> UZ_DEV int uz_en_h(typename En<LDS>::W w) { return (int)((w >> En<LDS>::SEQ_B) & (((typename En<LDS>::W)1 << En<LDS>::H_B) - 1)); }
template <bool LDS> UZ_DEV int uz_en_fb(typename En<LDS>::W w) { return (int)((w >> (En<LDS>::SEQ_B + En<LDS>::H_B)) & 3); }
template <bool LDS> UZ_DEV int uz_en_cb(typename En<LDS>::W w) { return (int)((w >> (En<LDS>::SEQ_B + En<LDS>::H_B + 2)) & 255); }
template <bool LDS> UZ_DEV typename En<LDS>::W uz_en_alleles(typename En<LDS>::W w, int fb, int cb) {
    return w | ((typename En<LDS>::W)fb << (En<LDS>::SEQ_B + En<LDS>::H_B)) | ((typename En<LDS>::W)cb << (En<LDS>::SEQ_B + En<LDS>::H_B + 2));
}
// Claim ranks of the chaining levels (phase E).  A claim is (e: frontier element, j: index into its read_sites list, krel: index into the
// site_reads list of the site, target haplotype); the smallest (e, j, krel) wins a pair, winners are ordered by (target, e, j, krel).
//   HBM build: 20 + 12 + 20 + 1 bits of a 64-bit word;  arena build: 10 + 8 + 13 + 1 bits of a 32-bit word (the DNM is given up to the HBM
//   build when a field would not fit: uz_phase_dnm checks nI, the winner room, E and every read_sites length).
template <bool LDS> struct Rk {
    typedef unsigned long long T;
    static constexpr int EB = 20, JB = 12, KB = 20;
};
template <> struct Rk<true> {
    typedef uint32_t T;
    static constexpr int EB = 10, JB = 8, KB = 13;
};
template <bool LDS> UZ_DEV typename Rk<LDS>::T uz_rk_none() { return (typename Rk<LDS>::T) ~(typename Rk<LDS>::T)0; }
// b = (e << 12) | j as site_best carries it
template <bool LDS> UZ_DEV typename Rk<LDS>::T uz_rk_make(unsigned long long b, int krel, int target) {
    typedef typename Rk<LDS>::T T;
    const T e = (T)(b >> 12), j = (T)(b & 0xFFFu);
    return (((((e << Rk<LDS>::JB) | j) << Rk<LDS>::KB) | (T)krel) << 1) | (T)target;
}
// winner key: target in the top bit, the rank below it
template <bool LDS> UZ_DEV typename Rk<LDS>::T uz_rk_win(typename Rk<LDS>::T k) {
    typedef typename Rk<LDS>::T T;
    return ((k & (T)1) << (8 * sizeof(T) - 1)) | (k >> 1);
}

// The working arrays of one DNM, listed once: X(element type, name, capacity in elements).  cA .. cFR are the capacities of the scratch
// layout (uz_scratch_layout); the element types hidx / pidx / xidx / rlen / gflg / skey / rkey are those of the build (ScrTy).
// ---- arrays the LDS build places in its arena (uz_scr_make points them at the arena before any request)
#define UZ_SCR_ARENA(X)                                                                                                                   \
    X(uint8_t, a_cls, cA) X(int32_t, a_flag0, cA) X(int32_t, a_flag1, cA) X(int32_t, LR, 2 * cA) X(int32_t, LA, 2 * cA) /* flags + lists: SV evidence only */ \
    X(int32_t, hpos, cH) X(int32_t, hcanon, cH) X(int32_t, h_a, cH) X(int32_t, h_off, cH) X(int32_t, sr_off, cH) X(uint8_t, sr_exists, cH) \
    X(int32_t, ov0, cH) /* enumerate index of the first record of every het-site fetch (the cut-off of :178-179) */                        \
    X(uint8_t, href, cH) X(uint8_t, halt, cH) /* REF / ALT base of every het site of the DNM */                                            \
    X(unsigned long long, site_best, cH) /* chaining: first frontier element finding an allele at a het index: (e << 12 | j) << 16 | allele << 8 | haplotype */ \
    X(int32_t, cpos, cC) X(uint32_t, cvote, cC) X(uint8_t, cflag, cC) X(uint8_t, cref, cC) X(uint8_t, calt, cC) /* per candidate: UZ_CF_* flags, REF and ALT base */ \
    X(int32_t, i_seg, cI) X(pidx, i_pair, cI) X(uint8_t, i_hb, cI) /* i_hb: bit 0 haplotype ("alt" list), bit 1 the element has a mate */  \
    X(uint32_t, i_q, cI) X(int32_t, i_soff, cI) X(uint32_t, i_pk, cI) /* (seeding step only) name id; first seed entry; matches qp | L << 8 | (R + 1) << 16 (arena build) */ \
    X(skey, keys, cM) /* pair table: (name, sequence) keys; after the sort the entry words (En) */                                         \
    X(xidx, reg_t, cT) /* (B .. P) per registration: the item of the fetch list it came from (its het site and record follow from h_off / h_a) */ \
    X(fidx, x16, cM)   /* (P .. D) per entry: its own record, then its pair's primary segment (1 + record - rbase, 0 = none) */             \
    X(hidx, seq_h, cM) /* het index of every seed entry */                                                                                 \
    X(xidx, rs_off, cM) X(rlen, rs_len, cM) X(gflg, grp, cM) X(gflg, pvote, cM)                                                            \
    X(fidx, pf0, cM) /* per pair: its primary segment -- fetched_reads[name][0], "last writer wins" (quirk Q11) -- as 1 + record - rbase, 0 = none */ \
    X(uint32_t, f_item, 2 * cM) /* join: the (pair, haplotype) items of the grouped pairs, compacted */                                      \
    X(rkey, win, cFR) X(pidx, w_pair, cFR) X(hidx, w_pos, cFR) /* winners of a chaining level: key, pair, canonical het index of the site */ \
    X(pidx, fr_pair, cFR) X(hidx, fr_pos, cFR) X(uint8_t, fr_hap, cFR) /* the frontier: pair, canonical het index of the site the element was claimed at (-1: an init element), haplotype */ \
    X(int32_t, misc, 8) /* [0] KeyError seen, [1] match_info count, [2] capacity exceeded */
// ---- arrays that stay in the HBM scratch in both builds (none of them on the point-variant path of the arena build: since round 5 a pair's
// records follow from arena arrays alone, and the second record of a pair is the first one's mate)
#define UZ_SCR_HBM_BOTH(X)                                                                                                                \
    X(int32_t, i_qp, cI) X(int32_t, i_L, cI) X(int32_t, i_R, cI) /* SV evidence: banned names, filter flags; HBM build: the seeding matches */ \
    X(uint32_t, pq, cM)            /* name id of every pair (the optional lists) */
// ---- ... and those only the HBM build touches
#define UZ_SCR_HBM_ONLY(X)                                                                                                                \
    X(unsigned long long, key, cM) /* second buffer of the counting sort; the winners' sorted copy of a chaining level */                  \
    X(int32_t, q_cnt, 2 * cM + 1026) X(int32_t, q_fill, 2 * cM + 1026) /* counting sort of the pair-table keys over the query-name id range */
#define UZ_SCR_HBM(X) UZ_SCR_HBM_BOTH(X) UZ_SCR_HBM_ONLY(X)

template <bool LDS>
struct ScrT {
    typedef typename ScrTy<LDS>::hidx hidx; // het index of the DNM (-1 = none)
    typedef typename ScrTy<LDS>::pidx pidx; // pair id
    typedef typename ScrTy<LDS>::xidx xidx; // index into the sorted pair-table entries
    typedef typename ScrTy<LDS>::rlen rlen; // length of a read_sites list
    typedef typename ScrTy<LDS>::gflg gflg; // per-pair flag sets (haplotype groups, votes)
    typedef typename ScrTy<LDS>::rkey rkey; // claim rank of a chaining level (Rk)
    typedef typename ScrTy<LDS>::fidx fidx; // a record relative to the DNM's first one
    // pair-table sort key: (name id, sequence number).  HBM build: id << 24 | sequence in 64 bits.  Arena build: the ids a DNM meets lie
    // close together (they are handed out in file order), so (id - smallest id of the DNM) and the sequence number share 32 bits -- a DNM
    // whose ids do not fit is given up to the HBM build
    typedef typename ScrTy<LDS>::skey skey;
    // (no pointer arrays in this struct: a dynamically indexed member would pin it in private memory)
#define UZ_X(TY, NAME, CNT) TY *NAME;
    UZ_SCR_ARENA(UZ_X)
    UZ_SCR_HBM(UZ_X)
#undef UZ_X
};
typedef ScrT<false> Scr; // the layout of the HBM scratch region: every array at its place

// Where every array starts inside a workgroup's scratch region (bytes).  The host works the layout out once per launch
// (uz_scratch_layout) and hands it to the kernel with its arguments; the kernel adds an offset to its region's base at the phase that
// uses the array -- a scalar load and a scalar add -- instead of carving ~75 pointers at its start and keeping them (two scalar registers
// each) alive across the whole body: that, with the ~100 scalar registers of the arguments themselves, was 430 spilled scalar registers
// and a third of the body's vector instructions moving them in and out of vector-register lanes (round 3's ISA).
struct ScrOff {
#define UZ_X(TY, NAME, CNT) unsigned long long NAME;
    UZ_SCR_ARENA(UZ_X)
    UZ_SCR_HBM(UZ_X)
#undef UZ_X
};
// fills `o`, returns the bytes of one workgroup's region.  arena_build: the region of a workgroup of the ARENA build -- it holds nothing but the few
// arrays that build keeps in HBM (UZ_SCR_HBM_BOTH: ~4 bytes per pair-table entry of the largest DNM), where the HBM build's region holds every
// array (~110 bytes per entry: 3.3 MB per workgroup for an SV batch -- sized like that for every wave of the arena build, a config-5 batch
// outgrew the 8 GiB scratch budget and ran on half its waves)
UZ_HD size_t uz_scratch_layout(const Caps &c, ScrOff &o, bool arena_build = false) {
    typedef Scr::hidx hidx; typedef Scr::pidx pidx; typedef Scr::xidx xidx; typedef Scr::rlen rlen; typedef Scr::gflg gflg; typedef Scr::skey skey; typedef Scr::rkey rkey; typedef Scr::fidx fidx;
    const size_t cA = (size_t)c.A + 1, cT = (size_t)c.T + 1, cH = (size_t)c.H + 2, cC = (size_t)c.C + 1, cI = (size_t)c.I + 2, cM = (size_t)c.M + 2;
    const size_t cFR = (cM > cI ? cM : cI) + 1;
    size_t at = 0;
#define UZ_X(TY, NAME, CNT) at = (at + 15) & ~(size_t)15; o.NAME = at; at += (size_t)(CNT) * sizeof(TY);
#define UZ_X0(TY, NAME, CNT) o.NAME = 0;
    if (arena_build) {
        UZ_SCR_ARENA(UZ_X0)
        UZ_SCR_HBM_ONLY(UZ_X0)
        UZ_SCR_HBM_BOTH(UZ_X)
    } else {
        UZ_SCR_ARENA(UZ_X)
        UZ_SCR_HBM(UZ_X)
    }
#undef UZ_X0
#undef UZ_X
    return (at + 255) & ~(size_t)255;
}
// the arrays that stay in the HBM scratch, from the layout and the region's base: called at every phase boundary of the body (the pointers
// then live for a phase, not for the kernel)
template <bool LDS>
UZ_DEV void uz_scr_hbm(ScrT<LDS> &s, const ScrOff &o, uint8_t *base) {
#define UZ_X(TY, NAME, CNT) s.NAME = reinterpret_cast<decltype(s.NAME)>(base + o.NAME);
    UZ_SCR_HBM(UZ_X)
#undef UZ_X
}
// HBM build: the other arrays too, once per DNM
UZ_DEV void uz_scr_all(Scr &s, const ScrOff &o, uint8_t *base) {
#define UZ_X(TY, NAME, CNT) s.NAME = reinterpret_cast<decltype(s.NAME)>(base + o.NAME);
    UZ_SCR_ARENA(UZ_X)
    UZ_SCR_HBM(UZ_X)
#undef UZ_X
}

// Per-DNM placement of the working arrays.  The scratch region in HBM has room for the largest DNM of the
// batch, but a typical DNM needs a few tens of KB, and every phase boundary waits for its outstanding stores:
// in LDS that wait is ~100 cycles instead of a round trip to memory.  The per-DNM body is compiled twice:
//   LDS build  the arrays the phases read over and over are placed in the workgroup's LDS arena in the order
//              they are requested, with their ACTUAL sizes ("temporary" arrays live above the persistent ones
//              and are recycled at phase boundaries); a DNM whose arrays do not fit is given up (return 1)
//              before anything of it is published and queued for the
//   HBM build  where every array keeps its place in the HBM scratch.
// Two builds instead of one address-space-agnostic body: a pointer that may hold either kind of address costs
// a 64-bit flat access (two address registers, both memory counters waited on); a pointer that can only hold an
// LDS address is a 32-bit ds_* access -- a third fewer instructions over the whole body.
struct Arena { // two-ended: persistent arrays grow from the bottom, temporaries from the top
    uint8_t *base;
    int cap, pers, top;
    int fail; // a request did not fit: the DNM is handed to the other build of the kernel (see uz_phase_dnm)
#ifdef UZ_EMU_STATS
    int peak; // most bytes in use at once (scripts/phase_sizes.py: what the host's arena estimate is fitted to)
#endif
};
#ifdef UZ_EMU_STATS
#define UZ_AR_PEAK(ar) do { const int u__ = (ar).pers + ((ar).cap - (ar).top); if (u__ > (ar).peak) (ar).peak = u__; } while (0)
#else
#define UZ_AR_PEAK(ar) ((void)0)
#endif
// LDS build: EVERY array requested here lies in the arena (its pointer never holds anything but an LDS address, so
// the compiler addresses it with 32-bit ds_* instructions); a request that does not fit sets `fail` and the caller
// gives the DNM up before the array is touched.  HBM build: the arrays keep their place in the HBM scratch.
template <bool LDS, typename T>
UZ_DEV void ar_p(Arena &ar, T *&ptr, size_t n) { // persistent for the rest of the DNM
    if (!LDS) return;
    const int b = (int)((n * sizeof(T) + 15) & ~(size_t)15);
    int at = ar.pers;
    if (ar.pers + b <= ar.top) ar.pers += b; else { ar.fail = 1; at = 0; }
    UZ_AR_PEAK(ar);
    ptr = reinterpret_cast<T *>(ar.base + at);
}
template <bool LDS, typename T>
UZ_DEV void ar_t(Arena &ar, T *&ptr, size_t n) { // until the next ar_reset
    if (!LDS) return;
    const int b = (int)((n * sizeof(T) + 15) & ~(size_t)15);
    int at = 0;
    if (ar.pers + b <= ar.top) { ar.top -= b; at = ar.top; } else ar.fail = 1;
    UZ_AR_PEAK(ar);
    ptr = reinterpret_cast<T *>(ar.base + at);
}
UZ_DEV void ar_reset(Arena &ar) { ar.top = ar.cap; }
template <bool LDS, typename T>
UZ_DEV void ar_pop(Arena &ar, T *, size_t n) { // gives back the temporary requested LAST (same element type and count)
    if (LDS) ar.top += (int)((n * sizeof(T) + 15) & ~(size_t)15);
}

struct PhaseArgs {
    int32_t n;
    int32_t min_gt_qual, readlen, no_extended, read_goal, evidence_min_ratio, split_error_margin;
    double cutoff;           // concordant insert cutoff of the kid (read_collector.py:11-25) ...
    const double *cutoff_d;  // ... or one per DNM (cohort batches: DNMs of several kids, uz_phase_cohort); null = the scalar
    // sites + window lists
    const int32_t *spos;
    const uint8_t *sref, *salt;
    const int64_t *cand_off, *het_off;
    const int32_t *cand_idx, *het_idx;
    const uint8_t *cand_flags;
    // DNMs
    const int32_t *rcontig, *dstart, *dend;
    const uint8_t *dflags, *vartype;
    const uint32_t *allele_off;
    const uint8_t *alleles;
    RD R;
    // results
    int32_t *status, *counts, *origin, *evidence;
    // optional lists: bump-allocated from one pool
    int32_t want_lists;
    int32_t *pool;
    unsigned long long pool_cap;
    unsigned long long *pool_cursor;
    long long *list_start; // [n] start of this DNM's block in the pool, -1 = none / overflow
    int32_t *list_len;     // [6n] dad_reads, mom_reads, dad_sites, mom_sites, ref group, alt group
    // scheduling + scratch
    int32_t *work_cursor;
    int32_t *retry_count; // DNMs this launch of the arena build gave up ...
    int32_t *retry_list;  // ... and their indices: the work list of the launch behind it
    // a launch that works through such a list instead of the whole batch (the second arena launch, with larger arenas; the HBM build)
    int32_t from_list, cursor_slot;
    const int32_t *src_count, *src_list;
    uint8_t *scratch;
    unsigned long long scratch_per_wg;
    Caps caps;
    ScrOff so; // where every working array starts inside a workgroup's scratch region (uz_scratch_layout)
    int32_t lds_arena_bytes;
    // fetch ranges found by the sizing pass (one lane per DNM, all DNMs in flight at once, so the
    // binary searches overlap instead of serialising inside the per-DNM workgroup)
    int32_t *pre_win; // [4n]   first / one-past-last record of the DNM fetch (SVs: of both breakpoint fetches)
    int32_t *pre_ha;  // [n_het] first record of every het-site fetch range
    int32_t *pre_hl;  // [n_het] its length
    unsigned long long *timing; // diagnostic builds only
};

// ------------------------------------------------------------------ helpers
UZ_DEV double uz_cutoff(const PhaseArgs &a, int d) { return a.cutoff_d ? a.cutoff_d[d] : a.cutoff; }

UZ_DEV long long uz_lower_bound(const int32_t *a, long long lo, long long hi, long long v) {
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if ((long long)a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// the same over the start field of the record headers
UZ_DEV long long uz_lower_bound_start(const RecA *ra, long long lo, long long hi, long long v) {
    while (lo < hi) {
        const long long mid = lo + ((hi - lo) >> 1);
        if ((long long)ra[mid].start < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Lower bounds on the start column, N of them side by side: the searches advance in lockstep, every step's N loads issued before the first of
// them is looked at -- a binary search is a chain of dependent loads at ~1 us each under load.  A plain binary search over a 1 GB column is ~25
// dependent HBM misses; the index levels keep all but a handful in L2 / L1.
// (Round 5: a DNM's window first and its 22 ranges inside it, side by side: 0.40 -> 0.38 ms per 100 k DNMs.  Round 6 tried interpolation search
// inside the window -- the first probes along the line between the bracket's ends, every other one from the fourth on a bisection: three or four
// probes per search on a pile-up's evenly spaced starts, but the chains of a wavefront advance in lockstep, 640 of them, and the wave takes as
// many steps as its unluckiest chain: close to bisection's eleven again; and the probe's arithmetic took the kernel from 120 to 142 registers and
// from 1 600 to 5 000 instructions: 0.37 -> 0.73 ms.  What worked instead is below: a second index level and no window.)
// N lower bounds over a whole contig's records [clo, chi) on three levels: the coarse index (every 4096th start:
// 180 KB for the bench's table, L2-resident -- ~12 steps), the mid index (every 64th start: the 64 entries under a coarse cell are two cache
// lines -- 6 steps, one or two misses) and the 64 record headers under a mid cell (1 KB -- 6 steps, about three misses).  Round 6: the twelve
// last steps used to go over the record headers alone, every one a 16-byte header in a line of its own -- 12 dependent misses per search.
// Chains with on[s] false take no step (their loads read the contig's first record) and return clo.
// (record indices and positions are below 2^31 -- abi.hip refuses larger tables -- so the chains keep 32-bit state: half the registers of a
// 64-bit version, twice the waves per SIMD: the pass is bound by the length of its chains of dependent loads times the rounds of waves the
// chip needs for a batch)
// One level: N lower bounds over ld(k), k in [a[s], b[s]), until every bracket is empty.
// (Tried in round 6 and dropped: by QUARTERS -- three probes per chain and step, issued together: a level of 64 entries in three dependent
// loads instead of six, the coarse index in six instead of twelve.  0.22 -> 0.26 ms per 100 k DNMs: the pass pays for every probe of every
// lane about two cycles of its CU -- 620 -> ~1 000 probes per DNM cost more than twelve steps less of the chain gave back.)
template <int N, typename F>
UZ_DEV void uz_lb_level(F ld, int32_t (&a)[N], int32_t (&b)[N], const int32_t (&v)[N], int32_t idle) {
    for (;;) {
        bool any = false;
        int32_t x[N], mid[N];
#pragma unroll
        for (int s = 0; s < N; s++) {
            const bool act = a[s] < b[s];
            any |= act;
            mid[s] = act ? a[s] + ((b[s] - a[s]) >> 1) : idle; // (an idle chain reads one entry again: no branch around the load)
            x[s] = ld(mid[s]);
        }
        if (!any) break;
#pragma unroll
        for (int s = 0; s < N; s++)
            if (a[s] < b[s]) { if (x[s] < v[s]) a[s] = mid[s] + 1; else b[s] = mid[s]; }
    }
}
UZ_DEV int32_t uz_clamp_i32(long long v) { return v < -0x7FFFFFFFLL ? -0x7FFFFFFF : (v > 0x7FFFFFFFLL ? 0x7FFFFFFF : (int32_t)v); } // (starts are int32: a value beyond either end compares like the end)
// The third level: a bracket [x, y) of records -- at most one 64-record cell, normally -- cut down to the eight records under one entry of mid8.
// The last steps of a search used to probe the record headers of its 64-record cell: a 16-byte header per probe, about four lines of the eight
// the cell spans -- and the sizing pass, 22 searches per DNM, was bound by that HBM traffic (round 6: 0.17 of its 0.22 ms).  The cell's eight
// entries are one 32-byte load; the eight headers under an entry are one line.
template <int N>
UZ_DEV void uz_mid8_refine(const int32_t *mid8, int32_t (&x)[N], int32_t (&y)[N], const int32_t (&v)[N]) {
    int32_t e[N][8], cell[N];
#pragma unroll
    for (int s = 0; s < N; s++) {
        cell[s] = x[s] >> 6;
        const int32_t *p = static_cast<const int32_t *>(__builtin_assume_aligned(mid8 + ((size_t)cell[s] << 3), 32)); // (the array is aligned, a cell is eight entries: two 16-byte loads)
#pragma unroll
        for (int j = 0; j < 8; j++) e[s][j] = p[j];
    }
#pragma unroll
    for (int s = 0; s < N; s++) {
        int32_t nx = x[s], ny = y[s];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int32_t r = (cell[s] << 6) + 8 * j;
            const bool valid = r > x[s] && r < y[s]; // (entries outside the bracket -- or beyond the table: never written -- decide nothing)
            if (valid && e[s][j] < v[s]) nx = r > nx ? r : nx;
            if (valid && !(e[s][j] < v[s])) ny = r < ny ? r : ny;
        }
        x[s] = nx; y[s] = ny;
    }
}
template <int N>
UZ_DEV void uz_lower_bounds_c(const RD &R, long long clo_, long long chi_, const long long (&v_)[N], const bool (&on)[N], long long (&res)[N]) {
    const int32_t clo = (int32_t)clo_, chi = (int32_t)chi_;
    int32_t lo[N], hi[N], v[N];
#pragma unroll
    for (int s = 0; s < N; s++) {
        lo[s] = clo; hi[s] = on[s] ? chi : clo;
        v[s] = uz_clamp_i32(v_[s]);
    }
    if (clo >= chi) {
#pragma unroll
        for (int s = 0; s < N; s++) res[s] = clo;
        return;
    }
    if (R.coarse && chi - clo > 8192) {
        const int32_t kl = (clo + 4095) >> 12, kh = chi >> 12;
        int32_t a[N], b[N];
#pragma unroll
        for (int s = 0; s < N; s++) { a[s] = kl; b[s] = (on[s] && kh > kl) ? kh : kl; }
        const int32_t *cx = R.coarse;
        uz_lb_level<N>([cx](int32_t k) { return cx[k]; }, a, b, v, kl);
#pragma unroll
        for (int s = 0; s < N; s++)
            if (on[s] && kh > kl) { lo[s] = a[s] > kl ? ((a[s] - 1) << 12) : clo; hi[s] = a[s] < kh ? (a[s] << 12) : chi; }
    }
    if (R.mid && chi - clo > 128) { // every chain inside its own stretch of the mid index
        int32_t a[N], b[N], kl[N], kh[N];
#pragma unroll
        for (int s = 0; s < N; s++) { kl[s] = (lo[s] + 63) >> 6; kh[s] = hi[s] >> 6; a[s] = kl[s]; b[s] = kh[s] > kl[s] ? kh[s] : kl[s]; }
        const int32_t *mx = R.mid;
        uz_lb_level<N>([mx](int32_t k) { return mx[k]; }, a, b, v, clo >> 6);
#pragma unroll
        for (int s = 0; s < N; s++)
            if (kh[s] > kl[s]) {
                const int32_t nlo = a[s] > kl[s] ? ((a[s] - 1) << 6) : lo[s], nhi = a[s] < kh[s] ? (a[s] << 6) : hi[s];
                lo[s] = nlo; hi[s] = nhi;
            }
    }
    if (R.mid8) uz_mid8_refine<N>(R.mid8, lo, hi, v);
    const RecA *rx = R.ra; // the last steps: every chain inside its own stretch of the column
    uz_lb_level<N>([rx](int32_t k) { return rx[k].start; }, lo, hi, v, clo);
#pragma unroll
    for (int s = 0; s < N; s++) res[s] = lo[s];
}
template <int N>
UZ_DEV void uz_lower_bounds_c(const RD &R, long long clo, long long chi, const long long (&v)[N], long long (&res)[N]) {
    bool on[N];
#pragma unroll
    for (int s = 0; s < N; s++) on[s] = true;
    uz_lower_bounds_c<N>(R, clo, chi, v, on, res);
}

// (pysam fetch(contig, lo, hi): candidates are records with start in [lo - max_span, hi); the caller still tests end > lo)

// index of `pos` in get_reference_positions(full_length=True), -1 if absent
// The fixed-width fields a base lookup needs, fetched together (one memory round trip) before the
// CIGAR walk instead of one by one along it.
struct SegHdr {
    int32_t start, end, n_cigar, l_seq;
    uint32_t cigar_off, sq_off, umask;
    int32_t mate;
};
// a record's rows: first staged unit + which units were staged
struct RowRef { uint32_t off, umask; };
UZ_DEV SegHdr uz_hdr(const RD &R, int seg) {
    const RecA A = R.ra[seg];
    const RecB B = R.rb[seg];
    SegHdr h;
    h.start = A.start; h.end = A.end; h.n_cigar = B.n_cigar; h.l_seq = B.l_seq;
    h.cigar_off = A.cigar_off; h.sq_off = A.sq_off; h.umask = R.umask[seg]; h.mate = B.mate;
    return h;
}
// position of base k inside the staged units of a row: false when its unit stayed home
UZ_DEV bool uz_unit_of(uint32_t umask, int k, uint32_t &unit) {
    const uint32_t u = (uint32_t)k >> 5;
    if (umask == UZ_UMASK_ALL) { unit = u; return true; }
    if (u > 14u || !((umask >> u) & 1u)) return false;
    const uint32_t below = umask & ((1u << u) - 1u);
#ifdef UZ_EMU
    unit = (uint32_t)__builtin_popcount(below);
#else
    unit = (uint32_t)__popc(below);
#endif
    return true;
}
UZ_DEV int uz_qidx_h(const RD &R, const SegHdr &h, long long pos) {
    // one operation spanning the whole read and as many reference bases as read bases can only be M / = / X
    // (the only operations that consume both; l_seq > 1 because an operation without reference bases still
    // gives end = start + 1): the index follows from the header, no CIGAR word is fetched
    if (h.n_cigar == 1 && h.l_seq > 1 && h.end - h.start == h.l_seq)
        return (pos >= h.start && pos < h.end) ? (int)(pos - h.start) : -1;
    const uint32_t *c = R.cigar + h.cigar_off;
    long long r = h.start;
    int q = 0;
    for (int k = 0; k < h.n_cigar; k++) {
        const int op = c[k] & 15, l = (int)(c[k] >> 4);
        if (op == UZ_OP_M || op == UZ_OP_EQ || op == UZ_OP_X) {
            if (pos >= r && pos < r + l) return q + (int)(pos - r);
            q += l; r += l;
        } else if (op == UZ_OP_I || op == UZ_OP_S) q += l;
        else if (op == UZ_OP_D || op == UZ_OP_N) r += l;
    }
    return -1;
}
UZ_DEV int uz_qidx(const RD &R, int seg, long long pos) { return uz_qidx_h(R, uz_hdr(R, seg), pos); }
UZ_DEV int uz_refpos_len(const RD &R, int seg) {
    const uint32_t *c = R.cigar + R.ra[seg].cigar_off;
    int q = 0;
    const int nc = R.rb[seg].n_cigar;
    for (int k = 0; k < nc; k++) {
        const int op = c[k] & 15, l = (int)(c[k] >> 4);
        if (op == UZ_OP_M || op == UZ_OP_EQ || op == UZ_OP_X || op == UZ_OP_I || op == UZ_OP_S) q += l;
    }
    return q;
}
UZ_DEV uint8_t uz_base(const RD &R, RowRef row, int k) {
    if (row.off == UZ_NO_SEQ_OFF) { *R.err = 1; return 0; } // the host left the bases out: its reach rule and the kernel disagree
    uint32_t u;
    if (!uz_unit_of(row.umask, k, u)) { *R.err = 3; return 0; } // ... or this 32-base unit of them
    const uint32_t code = uz_seq4_code(R.seq4, row.off + u, k & 31);
    // a record whose bases came as a list (UZ_UMASK_LISTED): a base nobody listed reads as code 0, which a listed base never is
    if (code == 0u && (row.umask & UZ_UMASK_LISTED) && row.umask != UZ_UMASK_ALL) { *R.err = 4; return 0; }
    return uz_nt16_ascii(code);
}
UZ_DEV bool uz_qual_low(const RD &R, RowRef row, int k) {
    if (row.off == UZ_NO_QLOW_OFF) { *R.err = 2; return false; } // a bit of a record that cannot be "good": the staging rule and the kernel disagree
    uint32_t u;
    if (!uz_unit_of(row.umask, k, u)) { *R.err = 3; return false; }
    return uz_qlow_bit(R.qlow, row.off + u, k & 31) != 0;
}

// binary_search (site_searcher.py:6-47): the result list is [qp, qp+1..R, qp-1..L]; returns its length
UZ_DEV int uz_bsearch(long long start, long long end, const int32_t *pos, int n, int &qp_out, int &L, int &Rr) {
    int qs = 0, qe = n - 1, qsp = -1, qep = -1;
    while (qe > -1) {
        if (qs > qe) break;
        if (qs == qsp && qe == qep) break;
        qsp = qs; qep = qe;
        const int qp = (qe + qs) / 2;
        const long long p = pos[qp];
        if (start <= p && p < end) {
            int r = qp, l = qp;
            while (r + 1 < n && start <= pos[r + 1] && pos[r + 1] <= end) r++;
            while (l - 1 >= 0 && start <= pos[l - 1] && pos[l - 1] <= end) l--;
            qp_out = qp; L = l; Rr = r;
            return r - l + 1;
        } else if (p > start) qe = qp - 1;
        else if (p < start) qs = qp + 1;
    }
    qp_out = 0; L = 0; Rr = -1;
    return 0;
}
UZ_DEV int uz_bsearch_nth(int m, int qp, int L, int Rr) {
    if (m == 0) return qp;
    if (m <= Rr - qp) return qp + m;
    return qp - (m - (Rr - qp));
}

// the filters shared by the DNM fetch and the het-site fetch (:395-418 / :181-214), without the
// site-specific parts.  Returns the mate or -1.
// All fields of the record are requested together, then all fields of its mate (two memory round
// trips instead of one per test); the tests keep the reference's order.
// the same on values already fetched: the record's headers and QC byte, its mate's first header and QC byte
UZ_DEV int uz_pair_ok_vals(const PhaseArgs &a, double cutoff, const RecA &A, const RecB &B, uint32_t qc, const RecA &M, uint32_t qm) {
    const int mate = B.mate;
    long long ins = (long long)B.tlen - 2LL * a.readlen;
    const long long rs = A.start, re = A.end, ms = M.start, me = M.end;
    if (ins < 0) ins = -ins;
    if (!(qc & UZ_QC_GOOD) || (double)ins > cutoff) return -1;
    if (mate < 0) return -1;
    if (!(qm & UZ_QC_GOOD)) return -1;
    if (!(qc & UZ_QC_NONE5) || !(qm & UZ_QC_NONE5)) return -1;
    if ((ms <= rs && rs <= me) || (ms <= re && re <= me)) return -1; // mates overlap
    return mate;
}

// everything the filters and the allele look-up read of a record, requested together (one memory round trip)
struct RecFull { RecA A; RecB B; uint32_t qc, umask; };
UZ_DEV RecFull uz_rec_full(const RD &R, int seg) {
    RecFull r;
    r.A = R.ra[seg]; r.B = R.rb[seg]; r.qc = uz_qc_of(R.qs[seg], R.min_map_qual); r.umask = R.umask[seg];
    return r;
}
UZ_DEV SegHdr uz_hdr_of(const RecFull &r) {
    SegHdr h;
    h.start = r.A.start; h.end = r.A.end; h.n_cigar = r.B.n_cigar; h.l_seq = r.B.l_seq;
    h.cigar_off = r.A.cigar_off; h.sq_off = r.A.sq_off; h.umask = r.umask; h.mate = r.B.mate;
    return h;
}
// Phase A classification of one record fetched at the DNM: 0 none, 1 "ref", 2 "alt".  Two round trips (the record, its mate) before the
// first test instead of four (filters, then the headers of both again for the allele look-up).
UZ_DEV int uz_classify_dnm_read(const RD &R, const PhaseArgs &a, double cutoff, int seg, long long flo, long long position,
                                const uint8_t *ref, int ref_len, const uint8_t *alt, int alt_len) {
    const RecFull r = uz_rec_full(R, seg);
    const RecFull m = uz_rec_full(R, r.B.mate >= 0 ? r.B.mate : seg); // a safe index: unused without a mate
    if (!((long long)r.A.end > flo)) return 0;
    const int mate = uz_pair_ok_vals(a, cutoff, r.A, r.B, r.qc, m.A, m.qc);
    if (mate < 0) return 0;
    const SegHdr hs = uz_hdr_of(r);
    if (ref_len == alt_len) { // snv_match_alleles :296-336; get_allele_at :56-73
        const SegHdr hm = uz_hdr_of(m);
        RowRef row = {0, UZ_UMASK_ALL};
        int at = -1;
        const int i = uz_qidx_h(R, hs, position);
        if (i >= 0) {
            if (i < 4 || i > a.readlen - 4) return 0;
            if (!(hs.l_seq > i + ref_len)) return 0; // the mate is not consulted (quirk Q10)
            row.off = hs.sq_off; row.umask = hs.umask; at = i;
        } else {
            const int j = uz_qidx_h(R, hm, position);
            if (j < 0 || j < 4 || j > a.readlen - 4 || !(hm.l_seq > j + ref_len)) return 0;
            row.off = hm.sq_off; row.umask = hm.umask; at = j;
        }
        bool eq = true;
        for (int k = 0; k < ref_len; k++) eq &= uz_base(R, row, at + k) == ref[k];
        if (eq) return 1;
        eq = true;
        for (int k = 0; k < alt_len; k++) eq &= uz_base(R, row, at + k) == alt[k];
        return eq ? 2 : 0;
    }
    // indel_match_alleles :266-293
    const int var_len = ref_len > alt_len ? ref_len : alt_len;
    const int rp = uz_qidx_h(R, hs, position);
    if (rp < 0) return 0;
    const uint32_t *c = R.cigar + hs.cigar_off;
    bool has_id = false;
    int oi = 0;
    const int nc = hs.n_cigar;
    for (int k = 0; k < nc && oi < rp + var_len; k++) { // per-op expansion indexed by the query index (quirk Q16)
        const int op = c[k] & 15, l = (int)(c[k] >> 4);
        const int a0 = oi > rp ? oi : rp, a1 = (oi + l) < (rp + var_len) ? (oi + l) : (rp + var_len);
        if (a0 < a1 && (op == UZ_OP_I || op == UZ_OP_D)) has_id = true;
        oi += l;
    }
    const int ls = hs.l_seq;
    const RowRef qrow = {R.qoff[seg], hs.umask};
    for (int k = rp; k < rp + var_len && k < ls; k++)
        if (uz_qual_low(R, qrow, k)) return 0;
    if (has_id) return 2;
    if (7 < rp && rp < uz_refpos_len(R, seg) - 7) return 1;
    return 0;
}

// collect_reads_sv (:499-586) for one record fetched around a breakpoint `position`:
// 0 nothing, 1 supporting [read, mate] (split read), 2 supporting [mate, read] (discordant pair or
// clipped read), 3 the record bans its query name (:520-522)
// What the rule reads of a record comes in two round trips -- uz_sv_fetch1: headers, QC word, flag word; uz_sv_fetch2: its mate's QC word
// and start, its first CIGAR word (98 % of records have one) -- which the caller issues for TWO records per lane before it decides either:
// a breakpoint fetch returns hundreds of records, and one wave works through them.
struct SvIn { RecA A; RecB B; uint32_t qc, fmw, qcm, w0; int32_t mstart; };
UZ_DEV void uz_sv_fetch1(const RD &R, int i, SvIn &x) { x.A = R.ra[i]; x.B = R.rb[i]; x.qc = uz_qc_of(R.qs[i], R.min_map_qual); x.fmw = R.fm[i]; }
UZ_DEV void uz_sv_fetch2(const RD &R, int i, SvIn &x) {
    const int mi = x.B.mate >= 0 ? x.B.mate : i; // a safe index: unused without a mate
    x.qcm = uz_qc_of(R.qs[mi], R.min_map_qual);
    x.mstart = R.ra[mi].start;
    x.w0 = x.B.n_cigar > 0 ? R.cigar[x.A.cigar_off] : 0u;
}
UZ_DEV int uz_sv_decide(const RD &R, const PhaseArgs &a, double cutoff, const SvIn &x, long long position, long long lo, long long sv_start,
                        long long sv_end) {
    const RecA &A = x.A;
    const RecB &B = x.B;
    if (!((long long)A.end > lo)) return 0;
    if (!(x.qc & UZ_QC_GOOD_DISC)) return 0;  // goodread(read, True) :503
    const int mate = B.mate;                  // :507-510
    if (mate < 0) return 0;
    if (!(x.qcm & UZ_QC_GOOD_DISC)) return 0; // :512
    const uint32_t *c = R.cigar + A.cigar_off;
    const int nc = B.n_cigar;
    auto cw = [&](int k) -> uint32_t { return k == 0 ? x.w0 : c[k]; };
    long long total = 0;
    for (int k = 0; k < nc; k++) total += (long long)(cw(k) >> 4);
    int start_m = 0, end_m = 0, lead = 0, trail = 0;
    {
        long long o = 0;
        const long long t0 = total - 10 > 0 ? total - 10 : 0;
        bool in_lead = true;
        for (int k = 0; k < nc; k++) { // M/= among the first / last 10 entries of the per-base expansion of all ops :515-519
            const uint32_t w = cw(k);
            const int op = w & 15;
            const long long l = (long long)(w >> 4), s0 = o, s1 = o + l;
            if (op == UZ_OP_M || op == UZ_OP_EQ) {
                const long long x1 = s1 < 10 ? s1 : 10;
                if (x1 > s0) start_m += (int)(x1 - s0);
                const long long y0 = s0 > t0 ? s0 : t0;
                if (s1 > y0) end_m += (int)(s1 - y0);
            }
            if (in_lead) {
                if (op == UZ_OP_S || op == UZ_OP_I) lead += (int)l;
                else if (op == UZ_OP_M || op == UZ_OP_EQ || op == UZ_OP_X) in_lead = false;
            }
            o = s1;
        }
        for (int k = nc - 1; k >= 0; k--) {
            const uint32_t w = cw(k);
            const int op = w & 15;
            if (op == UZ_OP_S || op == UZ_OP_I) trail += (int)(w >> 4);
            else if (op == UZ_OP_M || op == UZ_OP_EQ || op == UZ_OP_X) break;
        }
    }
    if (end_m < 7 && start_m < 7) return 3;
    const long long rs = A.start, re = A.end;
    if ((x.fmw >> 24) & UZ_AUX_HAS_SA) { // :524-533
        const long long m = a.split_error_margin;
        return ((position - m <= rs && rs <= position + m) || (position - m <= re && re <= position + m)) ? 1 : 0;
    }
    long long ins = (long long)B.tlen - 2LL * a.readlen;
    if (ins < 0) ins = -ins;
    const double var_len = (double)sv_end - (double)sv_start < 0 ? (double)sv_start - (double)sv_end : (double)sv_end - (double)sv_start;
    bool disc = (double)ins > cutoff;
    if (disc) {
        double ratio = var_len / (double)ins;
        if (ratio < 0) ratio = -ratio;
        disc = 0.7 < ratio && ratio < 1.3; // :534-536
    }
    if (disc) {
        const long long ms = x.mstart;
        const long long left0 = ms < rs ? ms : rs, right0 = ms > rs ? ms : rs;
        const long long wig = (long long)cutoff; // :551
        return ((sv_start - wig) < left0 && left0 < (sv_start + wig) && (sv_end - wig) < right0 && right0 < (sv_end + wig)) ? 2 : 0;
    }
    SegHdr hd; // (the headers are at hand: the index look-ups fetch nothing but CIGAR words, and those only of a multi-operation record)
    hd.start = A.start; hd.end = A.end; hd.n_cigar = B.n_cigar; hd.l_seq = B.l_seq; hd.cigar_off = A.cigar_off; hd.sq_off = A.sq_off; hd.umask = 0; hd.mate = B.mate;
    int rp = uz_qidx_h(R, hd, position); // :565-573
    if (rp < 0) rp = uz_qidx_h(R, hd, position - 1);
    if (rp < 0) rp = uz_qidx_h(R, hd, position + 1);
    if (rp < 0) return 0;
    int len = 0; // uz_refpos_len
    for (int k = 0; k < nc; k++) {
        const uint32_t w = cw(k);
        const int op = w & 15;
        if (op == UZ_OP_M || op == UZ_OP_EQ || op == UZ_OP_X || op == UZ_OP_I || op == UZ_OP_S) len += (int)(w >> 4);
    }
    if (rp < 2 || rp > len - 4) return 0;
    return (lead >= rp - 1 || trail >= len - (rp + 1)) ? 2 : 0; // :576-586
}

// ------------------------------------------------------------------ sizing
// Upper bounds of the per-DNM working set, computed before the scratch is sized:
//   b[0] records in the DNM fetch range, b[1] sum of the het-site fetch ranges, b[2] het sites,
//   b[3] candidates, b[4] max het sites inside any window of max_span+1 bases (bounds the
//   seeding matches of one init element).
// Lanes lane, lane + nlanes, ... share the het sites of DNM d; lane 0 also does the per-DNM part.
// The caller reduces (t_part: sum, mh_part: max) over the lanes and stores them as b[1], b[4].
// (Tried in round 4 and dropped: the window found by the 16 lanes of a DNM's group together, 16 probes per round -- seven rounds instead of ~24
// dependent loads.  0.40 -> 0.49 ms for 100 k DNMs: the first rounds of a 16-ary search land all over a 3 GB column, cold in every cache and
// TLB, where a binary search takes its first twelve steps on the 180 KB coarse index that stays in L2.)
// (Round 6: no window of the DNM first and the ranges searched inside it -- two dependent searches, ~23 dependent misses -- but every range straight
// over the contig on the three levels of uz_lower_bounds_c: ~12 L2 hits and ~4 misses per chain, the chains of a DNM side by side over its lanes.
// A point variant's own fetch is item 0 of the DNM's list, its het sites follow: no chain that is idle in every lane but one.)
// search(v, on, r): the 2 x UZ_BW_SITES lower bounds of a lane's items over the DNM's contig (uz_lower_bounds_c, or the sizing kernel's staged form)
template <typename SEARCH>
UZ_DEV void uz_phase_bounds_w(const PhaseArgs &a, int d, int32_t *b, int lane, int nlanes, long long &t_part, int &mh_part, SEARCH search) {
    const RD &R = a.R;
    const long long h0 = a.het_off[d];
    const int nh = (int)(a.het_off[d + 1] - h0);
    t_part = 0; mh_part = 0;
    const int tid = a.rcontig[d];
    const bool tid_ok = tid >= 0 && tid < R.n_contigs;
    const long long span = tid_ok ? R.max_span[tid] : 0;
    const long long clo = tid_ok ? R.contig_off[tid] : 0, chi = tid_ok ? R.contig_off[tid + 1] : 0;
    const bool point = a.vartype[d] == UZ_VT_POINT;
    if (lane == 0 && !point) { // collect_reads_sv fetches +-cutoff around both breakpoints (:478-497): four bounds over the contig, side by side
        long long fa = 0, fb = 0, fa2 = 0, fb2 = 0;
        if (tid_ok) {
            const long long icut = (long long)uz_cutoff(a, d);
            long long lo1 = (long long)a.dstart[d] - icut, lo2 = (long long)a.dend[d] - icut;
            if (lo1 < 0) lo1 = 0;
            if (lo2 < 0) lo2 = 0;
            // (two bounds at a time: four chains of three probes would set the register count of the whole pass)
            const long long v1[2] = {lo1 - span, (long long)a.dstart[d] + icut}, v2[2] = {lo2 - span, (long long)a.dend[d] + icut};
            long long r1[2], r2[2];
            uz_lower_bounds_c<2>(R, clo, chi, v1, r1);
            uz_lower_bounds_c<2>(R, clo, chi, v2, r2);
            fa = r1[0]; fb = r1[1] > r1[0] ? r1[1] : r1[0]; fa2 = r2[0]; fb2 = r2[1] > r2[0] ? r2[1] : r2[0];
        }
        b[0] = (int32_t)((fb - fa) + (fb2 - fa2));
        a.pre_win[4 * d] = (int32_t)fa; a.pre_win[4 * d + 1] = (int32_t)fb;
        a.pre_win[4 * d + 2] = (int32_t)fa2; a.pre_win[4 * d + 3] = (int32_t)fb2;
    }
    // The items of the DNM: a point variant's own fetch (item 0), then the fetch ranges of its het sites -- UZ_BW_SITES items per lane at a time, their
    // 2 x UZ_BW_SITES searches side by side.  The walks back to the first het site a record ending at a site could still reach go side by side too
    // (two dependent loads per step each).
    constexpr int SITES = UZ_BW_SITES, NCH = 2 * SITES;
    const int own = point ? 1 : 0;
    const int n_items = own + (a.no_extended ? 0 : nh);
    for (int t0 = lane; t0 < n_items; t0 += nlanes * SITES) {
        long long v[NCH], hp[SITES], r[NCH];
        bool on[NCH];
        int hs[SITES]; // het site of the item; -1: the DNM's own fetch; nh: no item
#pragma unroll
        for (int j = 0; j < SITES; j++) {
            const int t = t0 + j * nlanes;
            hs[j] = t < n_items ? t - own : nh;
            if (hs[j] < 0) {
                const long long position = a.dstart[d];
                const long long flo = (a.dflags[d] & UZ_DF_FETCH_FALLBACK) ? position : position - 1;
                hp[j] = position;
                v[2 * j] = flo - span; v[2 * j + 1] = position + 1;
            } else {
                hp[j] = hs[j] < nh ? (long long)a.spos[a.het_idx[h0 + hs[j]]] : 0;
                v[2 * j] = hp[j] - span; v[2 * j + 1] = hp[j] + 1;
            }
            on[2 * j] = on[2 * j + 1] = hs[j] < nh && tid_ok;
        }
        search(v, on, r);
        int left[SITES];
#pragma unroll
        for (int j = 0; j < SITES; j++) {
            left[j] = hs[j];
            const long long ra_ = tid_ok ? r[2 * j] : 0, rb_ = tid_ok ? (r[2 * j + 1] > r[2 * j] ? r[2 * j + 1] : r[2 * j]) : 0;
            if (hs[j] < 0) {
                b[0] = (int32_t)(rb_ - ra_);
                a.pre_win[4 * d] = (int32_t)ra_; a.pre_win[4 * d + 1] = (int32_t)rb_;
                a.pre_win[4 * d + 2] = 0; a.pre_win[4 * d + 3] = 0;
            } else if (hs[j] < nh) {
                a.pre_ha[h0 + hs[j]] = (int32_t)ra_; a.pre_hl[h0 + hs[j]] = (int32_t)(rb_ - ra_);
                t_part += rb_ - ra_;
            }
        }
        for (;;) { // first het site (the list is sorted) a record ending at hp could still reach back to
            bool any = false;
            long long q[SITES];
#pragma unroll
            for (int j = 0; j < SITES; j++) {
                const bool act = hs[j] >= 0 && hs[j] < nh && left[j] > 0;
                q[j] = act ? (long long)a.spos[a.het_idx[h0 + left[j] - 1]] : 0;
            }
#pragma unroll
            for (int j = 0; j < SITES; j++)
                if (hs[j] >= 0 && hs[j] < nh && left[j] > 0 && q[j] >= hp[j] - span - 1) { left[j]--; any = true; }
            if (!any) break;
        }
#pragma unroll
        for (int j = 0; j < SITES; j++)
            if (hs[j] >= 0 && hs[j] < nh && hs[j] - left[j] + 1 > mh_part) mh_part = hs[j] - left[j] + 1;
    }
}
UZ_DEV void uz_phase_bounds(const PhaseArgs &a, int d, int32_t *b, int lane, int nlanes, long long &t_part, int &mh_part) {
    const int nc = (int)(a.cand_off[d + 1] - a.cand_off[d]), nh = (int)(a.het_off[d + 1] - a.het_off[d]);
    t_part = 0; mh_part = 0;
    if (lane == 0) {
        b[0] = b[1] = b[4] = 0;
        b[2] = nh; b[3] = nc;
        for (int k = 0; k < 4; k++) a.pre_win[4 * d + k] = 0;
    }
    if (nc <= 0) return;
    const RD &R = a.R;
    const int tid = a.rcontig[d];
    const bool tid_ok = tid >= 0 && tid < R.n_contigs;
    const long long clo = tid_ok ? R.contig_off[tid] : 0, chi = tid_ok ? R.contig_off[tid + 1] : 0;
    constexpr int NCH = 2 * UZ_BW_SITES;
    uz_phase_bounds_w(a, d, b, lane, nlanes, t_part, mh_part,
                      [&R, clo, chi](const long long (&v)[NCH], const bool (&on)[NCH], long long (&r)[NCH]) { uz_lower_bounds_c<NCH>(R, clo, chi, v, on, r); });
}

// ------------------------------------------------------------------ one DNM
// The arguments of the kernel, read where they are used.  A by-value kernel argument is loaded whole at the kernel's entry: PhaseArgs
// is ~120 scalar registers of pointers and sizes, all alive across a 16 k-instruction body that has ~100 to give -- the compiler parked
// them in vector-register lanes and fetched them back one v_readlane at a time (round 3: 430 spilled scalar registers, 2 150 v_readlane
// among 8 138 vector instructions).  The body therefore takes the ADDRESS of the arguments in the kernarg segment (constant address
// space: every read is a scalar load) and re-reads what a phase uses at the phase's start, through a copy of the address the compiler
// cannot see through -- so nothing loaded for one phase stays alive into the next.  Pointers come back typed as global memory (a
// pointer loaded from memory would otherwise be generic: flat_* instead of global_* accesses).
#ifdef UZ_EMU
typedef const PhaseArgs *PhaseArgsK;
UZ_DEV void uz_args_load(PhaseArgs &a, PhaseArgsK &ap) { a = *ap; }
#else
typedef const __attribute__((address_space(4))) PhaseArgs *PhaseArgsK;
template <typename T>
UZ_DEV T *uz_g(T *p) { return (T *)(__attribute__((address_space(1))) T *)p; }
UZ_DEV void uz_args_load(PhaseArgs &a, PhaseArgsK &ap) {
    asm volatile("" : "+s"(ap)); // a fresh copy of the address: loads through it cannot be merged with (or hoisted above) earlier ones
#define UZ_V(f) a.f = ap->f;
#define UZ_P(f) a.f = uz_g(ap->f);
    UZ_V(n) UZ_V(min_gt_qual) UZ_V(readlen) UZ_V(no_extended) UZ_V(read_goal) UZ_V(evidence_min_ratio) UZ_V(split_error_margin) UZ_V(cutoff)
    UZ_P(cutoff_d) UZ_P(spos) UZ_P(sref) UZ_P(salt) UZ_P(cand_off) UZ_P(het_off) UZ_P(cand_idx) UZ_P(het_idx) UZ_P(cand_flags)
    UZ_P(rcontig) UZ_P(dstart) UZ_P(dend) UZ_P(dflags) UZ_P(vartype) UZ_P(allele_off) UZ_P(alleles)
    UZ_P(R.ra) UZ_P(R.rb) UZ_P(R.fm) UZ_P(R.contig_off) UZ_P(R.max_span) UZ_V(R.n_contigs) UZ_P(R.cigar) UZ_P(R.seq4) UZ_P(R.qlow) UZ_P(R.qoff)
    UZ_P(R.nlow) UZ_P(R.umask) UZ_P(R.err) UZ_P(R.qs) UZ_V(R.min_map_qual) UZ_P(R.coarse) UZ_P(R.mid) UZ_P(R.mid8)
    UZ_P(status) UZ_P(counts) UZ_P(origin) UZ_P(evidence) UZ_V(want_lists) UZ_P(pool) UZ_V(pool_cap) UZ_P(pool_cursor) UZ_P(list_start) UZ_P(list_len)
    UZ_P(work_cursor) UZ_P(retry_count) UZ_P(retry_list) UZ_V(from_list) UZ_V(cursor_slot) UZ_P(src_count) UZ_P(src_list) UZ_P(scratch) UZ_V(scratch_per_wg)
    UZ_V(caps.A) UZ_V(caps.T) UZ_V(caps.H) UZ_V(caps.C) UZ_V(caps.I) UZ_V(caps.M) UZ_V(lds_arena_bytes)
    UZ_P(pre_win) UZ_P(pre_ha) UZ_P(pre_hl) UZ_P(timing)
#define UZ_X(TY, NAME, CNT) UZ_V(so.NAME)
    UZ_SCR_ARENA(UZ_X)
    UZ_SCR_HBM(UZ_X)
#undef UZ_X
#undef UZ_V
#undef UZ_P
}
#endif
// a phase boundary of the body: the arguments and the HBM scratch pointers of the next phase are read afresh
#define UZ_PHASE_ARGS() do { uz_args_load(a, ap); uz_scr_hbm<LDS>(s, a.so, scr_base); } while (0)

// The working arrays of one build.  HBM build: every array at its place in the scratch region.  LDS build: the arrays that stay in HBM
// come from the scratch layout (uz_scr_hbm, at every phase boundary); the arena arrays are pointed at the arena before any request, so
// that none of them ever holds an HBM address in that build (the compiler then proves the address space of every access).
template <bool LDS>
UZ_DEV ScrT<LDS> uz_scr_make(const ScrOff &o, uint8_t *scr_base, uint8_t *b) {
    ScrT<LDS> s;
    if constexpr (!LDS) uz_scr_all(s, o, scr_base);
    else {
#define UZ_X(TY, NAME, CNT) s.NAME = reinterpret_cast<decltype(s.NAME)>(b);
        UZ_SCR_ARENA(UZ_X)
#undef UZ_X
        uz_scr_hbm<true>(s, o, scr_base);
    }
    return s;
}

// returns 0 when the DNM is done (its status and results are written), 1 when the LDS build gives it up
// ap: the kernel's arguments where they lie (device: the kernarg segment); scr_base: this workgroup's scratch region
//
// ONE WAVE works through a DNM (wg.hpp): lists are walked in rounds of 64 items, the place of an item in a compacted list is a
// ballot rank (wg_rank), and nothing waits at a barrier.  Order of the steps: A (DNM reads -> init elements), C (seeding matches of the
// init elements: before B, so that the pair table's size is known when B fills its keys), B (registrations, written compacted with
// their names as sort keys), S (sort), P (pair ids: one pass over the sorted keys), D, E, F.
template <bool LDS, typename SH>
UZ_DEV int uz_phase_dnm(PhaseArgsK ap, uint8_t *scr_base, SH *sh, uint8_t *lds_arena, int d) {
    typedef typename ScrT<LDS>::hidx hidx;
    typedef typename ScrT<LDS>::pidx pidx;
    typedef typename ScrT<LDS>::xidx xidx;
    typedef typename ScrT<LDS>::rlen rlen;
    typedef typename ScrT<LDS>::skey skey;
    typedef typename ScrT<LDS>::rkey rkey;
    typedef typename ScrT<LDS>::fidx fidx;
    PhaseArgs a;
    uz_args_load(a, ap);
    const RD &R = a.R;
    ScrT<LDS> s = uz_scr_make<LDS>(a.so, scr_base, lds_arena);
    Arena ar = {lds_arena, LDS ? a.lds_arena_bytes : 0, 0, LDS ? a.lds_arena_bytes : 0, 0};
#ifdef UZ_EMU_STATS
    ar.peak = 0;
#endif
    const long long c0 = a.cand_off[d], h0 = a.het_off[d];
    const int nc = (int)(a.cand_off[d + 1] - c0), nh = (int)(a.het_off[d + 1] - h0);
    WG_SYNC();
    WG_T0 {
        a.status[d] = UZ_ST_NO_CAND;
        a.origin[d] = UZ_OR_NONE; a.evidence[d] = 0;
        for (int k = 0; k < 4; k++) a.counts[4 * d + k] = 0;
        if (a.want_lists) { a.list_start[d] = -1; for (int k = 0; k < 6; k++) a.list_len[6 * d + k] = 0; }
    }
    if (nc <= 0) return 0; // snv_phaser.py:254-262
    if (nh > a.caps.H || nc > a.caps.C) { WG_T0 a.status[d] = UZ_ST_CAPACITY; return 0; }
    if (LDS && nh > 127) return 1; // 8-bit het indices of the arena build
    ar_p<LDS>(ar, s.misc, 8);
    ar_p<LDS>(ar, s.cpos, nc + 1); ar_p<LDS>(ar, s.cvote, nc + 1);
    ar_p<LDS>(ar, s.cflag, nc + 1); ar_p<LDS>(ar, s.cref, nc + 1); ar_p<LDS>(ar, s.calt, nc + 1);
    ar_p<LDS>(ar, s.hpos, nh + 1); ar_p<LDS>(ar, s.hcanon, nh + 1); ar_p<LDS>(ar, s.h_a, nh + 1);
    ar_p<LDS>(ar, s.h_off, nh + 2); ar_p<LDS>(ar, s.sr_off, nh + 2); ar_p<LDS>(ar, s.sr_exists, nh + 1); ar_p<LDS>(ar, s.ov0, nh + 1);
    ar_p<LDS>(ar, s.href, nh + 1); ar_p<LDS>(ar, s.halt, nh + 1); ar_p<LDS>(ar, s.site_best, nh + 1);
    if (LDS && ar.fail) return 1;
    WG_T0 { s.misc[0] = 0; s.misc[1] = 0; s.misc[2] = 0; }
    (void)a.rcontig;
    const long long position = a.dstart[d];
    const double cutoff = uz_cutoff(a, d);
    const uint8_t *ref = a.alleles + a.allele_off[2 * d];
    const int ref_len = (int)(a.allele_off[2 * d + 1] - a.allele_off[2 * d]);
    const uint8_t *alt = a.alleles + a.allele_off[2 * d + 1];
    const int alt_len = (int)(a.allele_off[2 * d + 2] - a.allele_off[2 * d + 1]);
    const long long flo = (a.dflags[d] & UZ_DF_FETCH_FALLBACK) ? position : position - 1;

    UZ_TICK_INIT;
    WG_FOR(k, nc) {
        const int si = a.cand_idx[c0 + k];
        s.cpos[k] = a.spos[si]; s.cvote[k] = 0;
        s.cflag[k] = a.cand_flags[c0 + k]; s.cref[k] = a.sref[si]; s.calt[k] = a.salt[si];
    }
    WG_FOR(k, nh) {
        const int si = a.het_idx[h0 + k];
        s.hpos[k] = a.spos[si]; s.href[k] = a.sref[si]; s.halt[k] = a.salt[si];
    }

    // ---- A: DNM reads -> init elements in seeding order: the "ref" list, then the "alt" list (:226); each hit contributes read, mate
    UZ_PHASE_ARGS(); // (the site columns of the set-up above are not needed again)
    const bool is_sv = a.vartype[d] != UZ_VT_POINT;
    const long long fa = a.pre_win[4 * d], fb = a.pre_win[4 * d + 1];
    const long long fa2 = a.pre_win[4 * d + 2], fb2 = a.pre_win[4 * d + 3];
    const int n0 = (int)(fb - fa);
    const int nA = n0 + (int)(fb2 - fa2);
    if (nA > a.caps.A) { WG_T0 a.status[d] = UZ_ST_CAPACITY; return 0; }
    ar_t<LDS>(ar, s.a_cls, nA + 1);
    if (LDS && ar.fail) return 1;
    int nre = 0, nae = 0, nI = 0; // elements of the "ref" / "alt" lists, of both
    if (!is_sv) {
        int n_ref = 0, n_alt = 0;
        WG_ROUNDS(i, nA, act) {
            const int cl = act ? uz_classify_dnm_read(R, a, cutoff, (int)(fa + i), flo, position, ref, ref_len, alt, alt_len) : 0;
            if (act) s.a_cls[i] = (uint8_t)cl;
            n_ref += wg_count(cl == 1);
            n_alt += wg_count(cl == 2);
        }
        WG_SYNC();
        UZ_TICK(0); // A.classify
        nre = 2 * n_ref; nae = 2 * n_alt; nI = nre + nae;
        if (nI > a.caps.I) { WG_T0 a.status[d] = UZ_ST_CAPACITY; return 0; }
        ar_p<LDS>(ar, s.i_seg, (size_t)nI + 2); ar_p<LDS>(ar, s.i_hb, (size_t)nI + 2); ar_p<LDS>(ar, s.i_pair, (size_t)nI + 2);
        ar_p<LDS>(ar, s.i_q, (size_t)nI + 2); ar_p<LDS>(ar, s.i_soff, (size_t)nI + 2);
        if (LDS && ar.fail) return 1;
        int br = 0, ba = 0;
        WG_ROUNDS(i, nA, act) { // a hit's place in its list is its rank among the hits of its class
            const int cl = act ? (int)s.a_cls[i] : 0;
            const int kr = wg_rank(cl == 1, br), ka = wg_rank(cl == 2, ba);
            if (cl) {
                const int m = cl == 1 ? 2 * kr : nre + 2 * ka;
                const int seg = (int)(fa + i);
                s.i_seg[m] = seg;
                s.i_seg[m + 1] = R.rb[seg].mate;
                s.i_hb[m] = s.i_hb[m + 1] = (uint8_t)(cl == 2 ? 1 : 0);
            }
        }
        WG_SYNC();
    } else {
        // ---- A (SV): collect_reads_sv :476-596 around both breakpoints -> "alt" list only
        ar_t<LDS>(ar, s.a_flag0, nA + 1); ar_t<LDS>(ar, s.a_flag1, nA + 1);
        if (LDS && ar.fail) return 1;
        const long long sv_start = a.dstart[d], sv_end = a.dend[d];
        const long long icut = (long long)cutoff;
        for (int tb = wg_lane_opaque(); tb < nA; tb += 2 * WG_NT) { // two records per lane and round: their loads in flight together
            int tt[2], ii[2];
            SvIn xin[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                tt[u] = tb + u * WG_NT;
                const int t = tt[u] < nA ? tt[u] : tb;
                ii[u] = (int)(t >= n0 ? fa2 + (t - n0) : fa + t);
                uz_sv_fetch1(R, ii[u], xin[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) uz_sv_fetch2(R, ii[u], xin[u]);
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int t = tt[u];
                if (t >= nA) continue;
                const long long bp = t >= n0 ? sv_end : sv_start;
                long long lo = bp - icut;
                if (lo < 0) lo = 0;
                const int code = uz_sv_decide(R, a, cutoff, xin[u], bp, lo, sv_start, sv_end);
                s.a_cls[t] = (uint8_t)code;
                s.a_flag0[t] = code == 3;
            }
        }
        const int nban = wg_exscan(s.a_flag0, nA, sh);
        WG_FOR(t, nA) { // the banned names, in fetch order: (item, name)
            if (s.a_cls[t] == 3) {
                const int k = s.a_flag0[t];
                const int i = (int)(t >= n0 ? fa2 + (t - n0) : fa + t);
                s.i_qp[k] = t;
                s.i_L[k] = (int32_t)R.rb[i].qname;
            }
        }
        WG_SYNC();
        WG_FOR(t, nA) { // a record is skipped once an EARLIER record of the same breakpoint banned its name (:501-502)
            int code = s.a_cls[t];
            if (code == 1 || code == 2) {
                const int i = (int)(t >= n0 ? fa2 + (t - n0) : fa + t);
                const int32_t q = (int32_t)R.rb[i].qname;
                for (int k = 0; k < nban; k++) {
                    const int tb = s.i_qp[k];
                    if (tb < t && ((tb >= n0) == (t >= n0)) && s.i_L[k] == q) { code = 0; break; }
                }
            } else code = 0;
            s.a_cls[t] = (uint8_t)code;
            s.a_flag1[t] = code ? 2 : 0;
        }
        const int nsup = wg_exscan(s.a_flag1, nA, sh);
        ar_t<LDS>(ar, s.LR, (size_t)nsup + 2); ar_t<LDS>(ar, s.LA, (size_t)nsup + 2);
        if (LDS && ar.fail) return 1;
        WG_FOR(t, nA) {
            const int code = s.a_cls[t];
            if (code) {
                const int i = (int)(t >= n0 ? fa2 + (t - n0) : fa + t);
                const int k = s.a_flag1[t];
                const int m = R.rb[i].mate;
                s.LR[k] = code == 1 ? i : m;      // :532-533
                s.LR[k + 1] = code == 1 ? m : i;  // :562-563, :585-586
            }
        }
        WG_SYNC();
        WG_FOR(k, nsup) { // :588-591 filter by the names banned at the LAST breakpoint only
            const int32_t q = (int32_t)R.rb[s.LR[k]].qname;
            int keep = 1;
            for (int j = 0; j < nban; j++)
                if (s.i_qp[j] >= n0 && s.i_L[j] == q) { keep = 0; break; }
            s.i_R[k] = keep;
        }
        const int nfil = wg_exscan(s.i_R, nsup, sh);
        if (nfil >= 2) { // :594-595
            WG_FOR(k, nsup) {
                const bool last = k + 1 == nsup;
                const int nxt = last ? nfil : s.i_R[k + 1];
                if (nxt > s.i_R[k]) s.LA[s.i_R[k]] = s.LR[k];
            }
            nae = nfil;
        }
        WG_SYNC();
        UZ_TICK(0); // A.classify
        nI = nae;
        if (nI > a.caps.I) { WG_T0 a.status[d] = UZ_ST_CAPACITY; return 0; }
        ar_p<LDS>(ar, s.i_seg, (size_t)nI + 2); ar_p<LDS>(ar, s.i_hb, (size_t)nI + 2); ar_p<LDS>(ar, s.i_pair, (size_t)nI + 2);
        ar_p<LDS>(ar, s.i_q, (size_t)nI + 2); ar_p<LDS>(ar, s.i_soff, (size_t)nI + 2);
        if (LDS && ar.fail) return 1;
        WG_FOR(m, nI) { s.i_seg[m] = s.LA[m]; s.i_hb[m] = 1; }
        WG_SYNC();
    }
    if (LDS && nI >= (1 << Rk<true>::EB)) return 1; // (a frontier element's index is a field of the 32-bit claim rank)

    // ---- het-site set-up of B (before the seeding step: it clears sr_exists, which the seeding step may set)
    int T = 0;
    if (!a.no_extended) {
        WG_FOR(h, nh) {
            s.h_a[h] = a.pre_ha[h0 + h];
            s.h_off[h] = a.pre_hl[h0 + h];
            s.hcanon[h] = h; // fixed below
            s.sr_exists[h] = 0;
        }
        WG_SYNC();
        WG_FOR(h, nh) { // first het index with the same position (the list is sorted by position)
            int c = h;
            while (c > 0 && s.hpos[c - 1] == s.hpos[h]) c--;
            s.hcanon[h] = c;
        }
        T = wg_exscan(s.h_off, nh, sh);
        if (T > a.caps.T) { WG_T0 a.status[d] = UZ_ST_CAPACITY; return 0; }
        WG_T0 s.h_off[nh] = T;
        WG_SYNC();
    }
    // rbase: the first record any array of this DNM names -- the arena build keeps records as 16-bit distances from it
    int rbase = 0;
    if constexpr (LDS) {
        int lo = 0x7FFFFFFF, hi = -1;
        WG_FOR(h, nh) {
            if (!a.no_extended && s.h_off[h + 1] > s.h_off[h]) {
                const int f = s.h_a[h], l = f + (s.h_off[h + 1] - s.h_off[h]) - 1;
                lo = f < lo ? f : lo; hi = l > hi ? l : hi;
            }
        }
        WG_FOR(m, nI) { const int f = s.i_seg[m]; lo = f < lo ? f : lo; hi = f > hi ? f : hi; }
        int mn, mx;
        wg_minmax(lo, hi, mn, mx, sh);
        if (mx >= 0 && (long long)mx - (long long)mn >= 65534) return 1; // (uniform)
        rbase = mx >= 0 ? mn : 0;
    }
    // ---- init elements: name, mate, and (C, :226-249) the matches of every element among the het sites -- one round trip for all of it
    ar_reset(ar); // (the classes and lists of phase A are not read again)
    ar_t<LDS>(ar, s.i_q, (size_t)nI + 2); ar_t<LDS>(ar, s.i_soff, (size_t)nI + 2);
    if (LDS) ar_t<LDS>(ar, s.i_pk, (size_t)nI + 1);
    if (LDS && ar.fail) return 1;
    WG_FOR(m, nI) {
        const int seg = s.i_seg[m];
        const RecA A = R.ra[seg];
        const RecB B = R.rb[seg];
        s.i_q[m] = B.qname;
        if (B.mate >= 0) s.i_hb[m] |= 2;
        int nm = 0, qp = 0, L = 0, Rr = -1;
        if (!a.no_extended && B.mate >= 0) nm = uz_bsearch(A.start, A.end, s.hpos, nh, qp, L, Rr);
        s.i_soff[m] = nm;
        if constexpr (LDS) s.i_pk[m] = (uint32_t)qp | ((uint32_t)L << 8) | ((uint32_t)(Rr + 1) << 16);
        else { s.i_qp[m] = qp; s.i_L[m] = L; s.i_R[m] = Rr; }
    }
    WG_SYNC();
    int E = 0, S = 0, P = 0;
    bool exception = false;
    if (!a.no_extended) {
        S = wg_exscan(s.i_soff, nI, sh);
        WG_T0 s.i_soff[nI] = S;
        WG_SYNC();
    }
    // ---- pair-table keys (name, sequence).  Sequence numbers: registration k -> k, seed j -> T + j, the presence entry of init element
    // m (every grouped pair needs an id) -> T + S + m: the reference's time order, known before the registrations are counted.  In the key
    // array the seeds and presence entries come first (S + nI of them), the registrations behind them as B compacts them.
    const int Mcap = T + S + nI;
    if (Mcap >= (1 << 20) || Mcap > a.caps.M) { // rank-key field widths / scratch: loud, never silent
        WG_T0 a.status[d] = UZ_ST_CAPACITY;
        return 0;
    }
    if (LDS && Mcap >= (1 << En<true>::SEQ_B)) return 1; // 15-bit sequence numbers of the arena build
    if (!LDS && nh >= (1 << En<false>::H_B)) { WG_T0 a.status[d] = UZ_ST_CAPACITY; return 0; }
    {
        ar_p<LDS>(ar, s.seq_h, (size_t)S + 1);
        size_t kcap = (size_t)Mcap + 2; // (a table beyond 1024 entries is sorted in place: room for the next power of two)
        if (LDS && Mcap > 1024) { kcap = 2048; while (kcap < (size_t)Mcap) kcap <<= 1; kcap += 1; }
        ar_p<LDS>(ar, s.keys, kcap);
    }
    if (LDS && ar.fail) return 1;
    int lmin = 0x7FFFFFFF, lmax = -1; // range of the query-name ids met (as int: ids beyond 2^31 take the bitonic path / the HBM build)
    auto put_key = [&](int pos, int seq, uint32_t q) {
        if constexpr (LDS) s.keys[pos] = q; // (shifted below, once the smallest id is known; the sequence number follows from the place)
        else s.keys[pos] = ((unsigned long long)q << 24) | (unsigned long long)seq;
        lmin = (int)q < lmin ? (int)q : lmin;
        lmax = (int)q > lmax ? (int)q : lmax;
    };
    if (!a.no_extended && S > 0 && nh > 0) { WG_T0 s.sr_exists[s.hcanon[nh - 1]] = 1; } // stale loop variable, :242-243 (quirk Q13)
    WG_FOR(m, nI) {
        const uint32_t q = s.i_q[m];
        if (!a.no_extended) {
            const int o = s.i_soff[m], nm = s.i_soff[m + 1] - o;
            if (nm > 0) {
                int qp, L, Rr;
                if constexpr (LDS) { const uint32_t pk = s.i_pk[m]; qp = (int)(pk & 255u); L = (int)((pk >> 8) & 255u); Rr = (int)(pk >> 16) - 1; }
                else { qp = s.i_qp[m]; L = s.i_L[m]; Rr = s.i_R[m]; }
                for (int j = 0; j < nm; j++) {
                    s.seq_h[o + j] = (hidx)uz_bsearch_nth(j, qp, L, Rr);
                    put_key(o + j, T + o + j, q);
                }
            }
        }
        put_key(S + m, T + S + m, q);
    }
    WG_SYNC();
    UZ_TICK(1); // A.lists + C
    ar_reset(ar); // (name ids and matches of the init elements are in the keys now)
    ar_t<LDS>(ar, s.x16, (size_t)Mcap + 1); ar_t<LDS>(ar, s.reg_t, (size_t)T + 1);
    if (LDS && ar.fail) return 1;
    const int kreg = S + nI; // place of registration 0 in the key array
    auto het_of = [&](int t) { // the fetch an item of the fetch list belongs to: last h with h_off[h] <= t (empty fetches share an offset: the last of them holds t)
        int lo = 0, hi = nh;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s.h_off[mid] <= t) lo = mid; else hi = mid; }
        return lo;
    };
    if (!a.no_extended) {
        // ---- B: registration at every het site, in list order (group_reads_by_haplotype :165-222).  Items = the records of all het-site
        // fetches end to end; a record that passes is written at its rank among those that pass.
        // Two items per lane and round: the loads of both (record headers and QC word, then the mates' header and QC word) are in
        // flight together before either is used.
        const bool need_ei = (long long)T > (long long)a.read_goal + 1; // the enumerate cut-off (:178-179) can only bite when a fetch returns more than read_goal + 1 records
        int OV = 0;
        const int lane = wg_lane_opaque();
        for (int t0 = 0; t0 < T; t0 += 2 * WG_NT) {
            int tt[2], hh[2], sg[2];
            bool act[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                tt[u] = t0 + u * WG_NT + lane;
                act[u] = tt[u] < T;
                const int t = act[u] ? tt[u] : T - 1;
                hh[u] = het_of(t);
                sg[u] = s.h_a[hh[u]] + (t - s.h_off[hh[u]]);
            }
            RecA A[2], Mt[2];
            RecB B[2];
            uint32_t q[2], qm[2];
#pragma unroll
            for (int u = 0; u < 2; u++) { A[u] = R.ra[sg[u]]; B[u] = R.rb[sg[u]]; q[u] = uz_qc_of(R.qs[sg[u]], R.min_map_qual); }
#pragma unroll
            for (int u = 0; u < 2; u++) { const int mi = B[u].mate >= 0 ? B[u].mate : sg[u]; Mt[u] = R.ra[mi]; qm[u] = uz_qc_of(R.qs[mi], R.min_map_qual); }
#pragma unroll
            for (int u = 0; u < 2; u++) { // (in item order: the 64 items of u = 0 come before those of u = 1)
                const int t = tt[u], h = hh[u];
                const bool first = act[u] && t == s.h_off[h]; // the first record of a fetch
                // overlap test, pair filters (pure: evaluated for every overlapping record, the enumerate cut-off only masks them)
                const bool ov = act[u] && (long long)A[u].end > (long long)s.hpos[h];
                const int mate = uz_pair_ok_vals(a, cutoff, A[u], B[u], q[u], Mt[u], qm[u]);
                bool ok = ov && mate >= 0 && (q[u] & UZ_QC_NM5);
                if (need_ei) { // enumerate index of the fetch iterator = overlapping records of this fetch before this one
                    const int er = wg_rank(ov, OV);
                    if (first) s.ov0[h] = er;
                    WG_SYNC();
                    if (ok) ok = !((er - s.ov0[h]) > a.read_goal); // :179
                }
                const int k = wg_rank(ok, E);
                if (first) // site_reads range of het index h starts here; so do the (empty) ranges in front of it that share its offset
                    for (int h2 = h; h2 >= 0 && s.h_off[h2] == t; h2--) s.sr_off[h2] = k;
                if (ok) {
                    s.reg_t[k] = (xidx)t;
                    put_key(kreg + k, k, B[u].qname);
                }
            }
        }
        WG_SYNC();
        UZ_TICK(3); // B.overlap
        // site_reads range of het index h: entries [sr_off[h], sr_off[h+1]); fetches that start at the end of the item list are empty
        WG_FOR(h, nh + 1) { if (s.h_off[h] >= T) s.sr_off[h] = E; }
        WG_SYNC();
        WG_FOR(h, nh) {
            // a canonical site exists in site_reads once any of its duplicates registered a read (:217-218)
            if (s.sr_off[h + 1] > s.sr_off[h]) s.sr_exists[s.hcanon[h]] = 1;
        }
        WG_SYNC();
    }
    UZ_TICK(4); // B.finish
    const int M = E + S + nI;
    if (LDS && E >= (1 << Rk<true>::KB)) return 1; // (an index into a site_reads list is a field of the 32-bit claim rank)
    int SB = 24; // bits of a key that hold the sequence number
    if constexpr (LDS) { SB = 1; while ((1 << SB) < Mcap) SB++; }
    // Sort by (query-name id, sequence).
    int qmin = -1, qmax = -1;
    {
        wg_minmax(lmin, lmax, qmin, qmax, sh);
        if constexpr (LDS) {
            // the keys sit in the arena, packed into 32 bits: bitonic sort in registers (up to 16 keys per lane)
            if (qmin < 0 || (((unsigned)(qmax - qmin)) >> (32 - SB)) != 0u) return 1; // (uniform) the HBM build sorts 64-bit keys
            WG_FOR(x, M) s.keys[x] = ((s.keys[x] - (uint32_t)qmin) << SB) | (uint32_t)(x < kreg ? T + x : x - kreg);
            WG_SYNC();
            UZ_TICK(5); // S.keys
            wg_sort32_lds(s.keys, M);
        } else {
            // keys in HBM scratch.  Name ids are interned in file order, so the names met around one locus span a short
            // id range: a counting sort over that range (stable order inside a bucket restored by a tiny insertion
            // sort) replaces the stages of a bitonic sort; wider ranges fall back to it.  Same array.
            const int qrange = M > 0 ? qmax - qmin + 1 : 0;
            if (M > 1 && qmin >= 0 && qrange > 0 && qrange <= 2 * a.caps.M + 1024) {
                WG_FOR(i, qrange + 1) { s.q_cnt[i] = 0; s.q_fill[i] = 0; }
                WG_SYNC();
                WG_FOR(x, M) wg_atomic_add(&s.q_cnt[(int)(s.keys[x] >> 24) - qmin], 1);
                (void)wg_exscan(s.q_cnt, qrange, sh);
                WG_FOR(x, M) {
                    const int b = (int)(s.keys[x] >> 24) - qmin;
                    s.key[s.q_cnt[b] + wg_atomic_add(&s.q_fill[b], 1)] = s.keys[x];
                }
                WG_SYNC();
                WG_FOR(b, qrange) {
                    const int n_b = s.q_fill[b];
                    if (n_b > 1) {
                        unsigned long long *v = s.key + s.q_cnt[b];
                        for (int i = 1; i < n_b; i++) {
                            const unsigned long long kx = v[i];
                            int j = i - 1;
                            while (j >= 0 && v[j] > kx) { v[j + 1] = v[j]; j--; }
                            v[j + 1] = kx;
                        }
                    }
                }
                WG_SYNC();
                WG_FOR(x, M) s.keys[x] = s.key[x]; // (back into the entry array: `key` is scratch again -- the winners' sorted copy in phase E)
                WG_SYNC();
            } else
                wg_sort64(s.keys, M, sh);
        }
    }
    UZ_TICK(6); // S.sort
    // ---- P: pair ids.  A pair = a run of equal names in the sorted keys; its id = the number of run starts before it.
    const skey seqmask = (skey)(((skey)1 << SB) - 1);
    {
        uint32_t carry = 0;
        WG_ROUNDS(x, M, act) {
            const uint32_t nmx = act ? (uint32_t)(s.keys[x] >> SB) : 0u;
            const uint32_t pv = wg_prev32(nmx, carry);
            P += wg_count(act && (x == 0 || nmx != pv));
        }
    }
    ar_p<LDS>(ar, s.rs_off, (size_t)P + 2); // (the other per-pair arrays once the registrations' places have been read: their room is reg_t's)
    if (LDS && ar.fail) return 1;
    UZ_TICK(7); // P.count
    {
        // one pass over the sorted keys: run starts -> pair ids; every key becomes its entry word (sequence number, het index); x16 takes
        // the entry's own record
        uint32_t carry = 0;
        int pb = 0;
        WG_ROUNDS(x, M, act) {
            const skey key = act ? s.keys[x] : (skey)0;
            const uint32_t nmx = (uint32_t)(key >> SB);
            const uint32_t pv = wg_prev32(nmx, carry);
            const bool st = act && (x == 0 || nmx != pv);
            const int pid = wg_rank(st, pb) + (st ? 1 : 0) - 1; // run starts up to and including this entry, minus one
            if (act) {
                const int seq = (int)(key & seqmask);
                int h = 0, own = 0;
                if (seq < T) { // a registration: het site and record from its place in the fetch list
                    const int t = (int)s.reg_t[seq];
                    h = het_of(t);
                    own = s.h_a[h] + (t - s.h_off[h]) - rbase + 1;
                } else if (seq < T + S) h = s.seq_h[seq - T];
                else {
                    own = s.i_seg[seq - T - S] - rbase + 1;
                    s.i_pair[seq - T - S] = (pidx)pid;
                }
                if (st) {
                    s.rs_off[pid] = (xidx)x;
                    if (a.want_lists) s.pq[pid] = LDS ? nmx + (uint32_t)qmin : nmx; // name id of the pair (the optional lists)
                }
                s.keys[x] = uz_en_make<LDS>(seq, h);
                s.x16[x] = (fidx)own;
            }
        }
        WG_T0 s.rs_off[P] = (xidx)M;
        WG_SYNC();
    }
    ar_pop<LDS>(ar, s.reg_t, (size_t)T + 1);
    ar_p<LDS>(ar, s.rs_len, (size_t)P + 1); ar_p<LDS>(ar, s.grp, (size_t)P + 1); ar_p<LDS>(ar, s.pf0, (size_t)P + 1);
    if (LDS && ar.fail) return 1;
    UZ_TICK(8); // P.scatter
    WG_FOR(p, P) {
        const int x0 = s.rs_off[p], x1 = s.rs_off[p + 1];
        int len = 0, lastx = -1;
        for (int x = x0; x < x1; x++) { // ascending sequence = the reference's time order: the last writer of fetched_reads wins (quirk Q11)
            const int seq = uz_en_seq<LDS>(s.keys[x]);
            if (seq < T + S) len++;
            if (seq < T) lastx = x;                                           // :222
            else if (seq >= T + S && (s.i_hb[seq - T - S] & 2)) lastx = x;   // :233-234 (an element with a mate)
        }
        const fidx f = lastx >= 0 ? s.x16[lastx] : (fidx)0;
        for (int x = x0; x < x1; x++) s.x16[x] = f; // every entry of the pair now names the pair's primary segment
        if (len >= (1 << Rk<LDS>::JB)) s.misc[2] = 1; // rank key: the bits of the read_sites index
        s.rs_len[p] = (rlen)len;
        s.pf0[p] = f;
        s.grp[p] = 0;
    }
    WG_SYNC();
    WG_FOR(m, nI) wg_or_flag(s.grp, (int)s.i_pair[m], (s.i_hb[m] & 1) ? 2u : 1u); // :230
    WG_SYNC();
    if (s.misc[2]) {
        if (LDS) return 1; // (the HBM build's rank has room for 4095 sites per pair)
        WG_SYNC(); WG_T0 a.status[d] = UZ_ST_CAPACITY; return 0;
    }

#ifdef UZ_EMU_STATS
    uz_emu_stats[0] += E; uz_emu_stats[1] += S; uz_emu_stats[2] += nI; uz_emu_stats[3] += P; uz_emu_stats[7] += nh; uz_emu_stats[8] += nc; uz_emu_stats[9]++;
    uz_emu_stats[10] += T;
    if (uz_emu_log) { long long *r = uz_emu_log + 12 * (long long)d; r[0] = nc; r[1] = nh; r[2] = nA; r[3] = T; r[4] = nI; r[5] = E; r[6] = S; r[7] = M; r[8] = P; }
#endif
    UZ_TICK(9); // P.pairs
    // the primary segment of a pair -- fetched_reads[name][0], "last writer wins" (quirk Q11) -- or -1; the second one is its mate
    // (a registration's mate and an init element's mate both are the record's own mate field).  Arena arrays only.
    auto primary = [&](int p) -> int {
        const int f = (int)s.pf0[p];
        return f ? rbase + f - 1 : -1;
    };
    if (!a.no_extended) {
        // ---- D: static allele tables.  One pass over the sorted entries: the pair-level finder allele
        // at the entry's het site (get_allele_at, :91-105) and, for registrations, the base the
        // pair's primary segment shows there (:114-124).  Both start from the same index into the
        // primary segment, computed once; both go into the entry's word.
        // UZ_PHASE_K entries per lane and round, staged: (i) the entries' primary-segment headers and quality-plane offsets,
        // (ii) the mates' headers where the primary segment does not cover the site, (iii) the base / quality bit --
        // each stage's loads of all the lane's entries are in flight together (a DNM's ~400 entries: two rounds of three round trips).
        {
            constexpr int K = UZ_PHASE_K;
            const int lane = wg_lane_opaque();
            for (int xb = 0; xb < M; xb += K * WG_NT) {
                int xx[K], hh[K], qi[K], mt[K];
                skey wv[K];
                RowRef row[K];
                uint32_t q0[K], um0[K];
                bool live[K], own[K], needm[K], isreg[K];
                {
                    SegHdr hd0[K];
#pragma unroll
                    for (int u = 0; u < K; u++) {
                        xx[u] = xb + u * WG_NT + lane;
                        const int x = xx[u] < M ? xx[u] : 0;
                        wv[u] = s.keys[x];
                        const int seq = uz_en_seq<LDS>(wv[u]);
                        hh[u] = uz_en_h<LDS>(wv[u]);
                        isreg[u] = seq < T;
                        const int fx = (int)s.x16[x];
                        live[u] = xx[u] < M && seq < T + S && fx != 0;
                        const int f = live[u] ? rbase + fx - 1 : 0;
                        hd0[u] = uz_hdr(R, f);
                        q0[u] = R.qoff[f];
                    }
#pragma unroll
                    for (int u = 0; u < K; u++) { // the primary segment covers the site: the mate is not consulted (quirk Q10)
                        qi[u] = -1; row[u].off = 0; row[u].umask = UZ_UMASK_ALL; own[u] = false; needm[u] = false;
                        um0[u] = hd0[u].umask; mt[u] = hd0[u].mate;
                        if (!live[u]) continue;
                        const int i = uz_qidx_h(R, hd0[u], (long long)s.hpos[hh[u]]);
                        if (i >= 0) {
                            own[u] = true;
                            if (i >= 4 && i <= a.readlen - 4 && hd0[u].l_seq > i + 1) { qi[u] = i; row[u].off = hd0[u].sq_off; row[u].umask = hd0[u].umask; }
                        } else needm[u] = mt[u] >= 0;
                    }
                }
                {
                    SegHdr h1[K];
#pragma unroll
                    for (int u = 0; u < K; u++) h1[u] = uz_hdr(R, needm[u] ? mt[u] : 0);
#pragma unroll
                    for (int u = 0; u < K; u++) {
                        if (!needm[u]) continue;
                        const int j = uz_qidx_h(R, h1[u], (long long)s.hpos[hh[u]]);
                        if (j >= 4 && j <= a.readlen - 4 && h1[u].l_seq > j + 1) { qi[u] = j; row[u].off = h1[u].sq_off; row[u].umask = h1[u].umask; }
                    }
                }
                uint8_t al[K];
                bool low[K];
#pragma unroll
                for (int u = 0; u < K; u++) {
                    al[u] = qi[u] >= 0 ? uz_base(R, row[u], qi[u]) : (uint8_t)0;
                    low[u] = (qi[u] >= 0 && own[u] && isreg[u]) ? uz_qual_low(R, RowRef{q0[u], um0[u]}, qi[u]) : true; // (only a registration uses it, below)
                }
#pragma unroll
                for (int u = 0; u < K; u++) {
                    if (xx[u] >= M || qi[u] < 0) continue; // (the word was written without alleles)
                    const int h = hh[u];
                    const int fbc = al[u] == s.href[h] ? 1 : (al[u] == s.halt[h] ? 2 : 0); // :98-105
                    const int cb = (own[u] && isreg[u] && !low[u]) ? (int)al[u] : 0;        // :114-124
                    if (fbc | cb) s.keys[xx[u]] = uz_en_alleles<LDS>(wv[u], fbc, cb);
                }
            }
        }
        WG_SYNC();
        UZ_TICK(10); // D.finder
        // ---- E: chaining.  Level 0 visits new_reads "alt" then "ref" (:224); deeper levels "ref" then "alt" (:78)
        int F = nI;
        // winners of one level: at most P.  The arena build sets room aside for 96 (a level rarely has more than a few
        // dozen: 19 at the median of the bench workload, 80 at most) and gives the DNM up if a level overflows it.
        const int wcap = (LDS && P > 96) ? 96 : P;
        {
            ar_reset(ar); // (the entries' primary segments are not read again: the join takes them per pair)
            const size_t fr = (size_t)(wcap > nI ? wcap : nI) + 2;
            ar_t<LDS>(ar, s.fr_pair, fr); ar_t<LDS>(ar, s.fr_pos, fr); ar_t<LDS>(ar, s.fr_hap, fr);
            ar_t<LDS>(ar, s.win, (size_t)wcap + 2); ar_t<LDS>(ar, s.w_pair, (size_t)wcap + 2); ar_t<LDS>(ar, s.w_pos, (size_t)wcap + 2);
        }
        if (LDS && ar.fail) return 1;
        WG_FOR(e, nI) {
            const int na = nae;
            const int m = e < na ? (nre + e) : (e - na);
            s.fr_pair[e] = s.i_pair[m];
            s.fr_pos[e] = (hidx)-1;
            s.fr_hap[e] = (uint8_t)(s.i_hb[m] & 1);
        }
        // (a pair is "assigned" exactly when it carries a haplotype bit in grp: the init pairs from the start, the winners of a
        // level from its end)
        WG_FOR(h, nh) s.site_best[h] = ~0ULL;
        WG_SYNC();
        UZ_TICK(11); // E.setup
        while (F > 0) {
            // (i) per het index, the first frontier element (in visiting order e, then read_sites
            // index j) that finds a usable allele there.  Every element finding REF or ALT at a het
            // index claims the same entries of its site_reads list, so only that first one can win.
            WG_FOR(e, F) {
                const int p = s.fr_pair[e];
                const int fcanon = s.fr_pos[e]; // canonical het index of the site this element was claimed at
                const unsigned long long hap = s.fr_hap[e];
                const int x0 = s.rs_off[p], len = s.rs_len[p];
                for (int j = 0; j < len; j++) {
                    const skey w = s.keys[x0 + j];
                    const int h = uz_en_h<LDS>(w);
                    if (s.hcanon[h] == fcanon) continue;        // :89-90 (same position <=> same canonical index)
                    const int fbc = uz_en_fb<LDS>(w);
                    if (!fbc) continue;                          // :104-105
                    if (!s.sr_exists[s.hcanon[h]]) { s.misc[0] = 1; continue; } // :106 KeyError
                    const unsigned long long fbv = fbc == 1 ? s.href[h] : s.halt[h];
                    // the allele and haplotype of the finder ride along below the rank, so the
                    // registrations can use them without going back to the frontier arrays
                    wg_atomic_min64(&s.site_best[h], ((((unsigned long long)e << 12) | (unsigned long long)j) << 16) | (fbv << 8) | hap);
                }
            }
            WG_SYNC();
            // (ii) every still-unassigned pair looks up the finder(s) of the sites its registrations stand at; the smallest claim
            // (frontier element, its read_sites index, the registration's place in the site's list) takes the pair (:108-141)
            int W = 0;
            WG_ROUNDS(p, P, act) {
                rkey best = uz_rk_none<LDS>();
                int bcanon = 0;
                if (act && !s.grp[p]) { // :108-110 (assigned before this level)
                    const int x1 = s.rs_off[p + 1];
                    for (int x = s.rs_off[p]; x < x1; x++) {
                        const skey w = s.keys[x];
                        const int k = uz_en_seq<LDS>(w);
                        if (k >= T) break; // (the registrations of a pair come first: smallest sequence numbers)
                        const uint8_t cb = (uint8_t)uz_en_cb<LDS>(w);
                        if (!cb) continue;
                        const int canon = s.hcanon[uz_en_h<LDS>(w)];
                        const int krel = k - s.sr_off[canon];
                        for (int h = canon; h < nh && s.hpos[h] == s.hpos[canon]; h++) { // het indices sharing the position
                            const unsigned long long sb = s.site_best[h];
                            if (sb == ~0ULL) continue;
                            const unsigned long long b = sb >> 16; // (e << 12) | j
                            const uint8_t fbv = (uint8_t)(sb >> 8);
                            const uint8_t nonf = fbv == s.href[h] ? s.halt[h] : s.href[h];
                            const int hap = (int)(sb & 1ULL);
                            int target;
                            if (cb == fbv) target = hap;                 // :134-136
                            else if (cb == nonf) target = hap ^ 1;       // :137-141
                            else continue;
                            const rkey key = uz_rk_make<LDS>(b, krel, target);
                            if (key < best) { best = key; bcanon = canon; }
                        }
                    }
                }
                // the winners in any order: their place in the next frontier is decided below by the RANK of their key
                const bool won = best != uz_rk_none<LDS>();
                const int wi = wg_rank(won, W);
                if (won && wi < wcap) { s.win[wi] = uz_rk_win<LDS>(best); s.w_pair[wi] = (pidx)p; s.w_pos[wi] = (hidx)bcanon; }
            }
            WG_SYNC();
            if (LDS && W > wcap) return 1; // (uniform)
            UZ_TICK(12); // E.expand
            WG_FOR(h, nh) s.site_best[h] = ~0ULL; // for the next level
            // winners in the order the reference appends them: "ref" targets by rank, then "alt" targets by rank.  Position in the
            // next frontier = rank of the (target, rank) key among the winners: counted directly while a level has few winners (the
            // arena build always: at most 96), looked up in a sorted copy otherwise
            const bool by_count = LDS || W <= 96;
            if constexpr (!LDS) {
                if (!by_count) {
                    WG_FOR(w, W) s.key[w] = s.win[w];
                    WG_SYNC();
                    wg_sort64(s.key, W, sh, false);
                }
            }
            WG_FOR(w, W) {
                const rkey ok = s.win[w];
                int posn = 0;
                if (by_count) {
                    for (int v = 0; v < W; v++) posn += s.win[v] < ok;
                } else if constexpr (!LDS) {
                    int lo = 0, hi = W;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s.key[mid] < ok) lo = mid + 1; else hi = mid; }
                    posn = lo;
                }
                const int target = (int)(ok >> (8 * sizeof(rkey) - 1));
                const int p = s.w_pair[w];
                s.fr_pair[posn] = (pidx)p;
                s.fr_pos[posn] = s.w_pos[w];
                s.fr_hap[posn] = (uint8_t)target;
                // every winner is a different pair: mark it here
                wg_or_flag(s.grp, p, target ? 2u : 1u);
            }
            WG_SYNC();
#ifdef UZ_EMU_STATS
            uz_emu_stats[4]++; if (W > uz_emu_stats[5]) uz_emu_stats[5] = W; uz_emu_stats[6] += W;
            if (uz_emu_log) { long long *r = uz_emu_log + 12 * (long long)d; if (W > r[9]) r[9] = W; r[10]++; }
#endif
            UZ_TICK(14); // E.frontier
            F = W;
        }
        exception = s.misc[0] != 0;
    }
    WG_SYNC();
    if (exception) {
        WG_T0 a.status[d] = UZ_ST_REF_EXCEPTION;
        return 0;
    }

    // ---- F: join + vote.  Items: extended -> both fetched segments of every grouped pair per
    // haplotype (:254-263); --no-extended -> the init list elements themselves.
    ar_reset(ar);
    ar_t<LDS>(ar, s.pvote, (size_t)P + 1);
    if (LDS && ar.fail) return 1;
#ifdef UZ_EMU_STATS
    if (uz_emu_log) uz_emu_log[12 * (long long)d + 11] = ar.peak;
#endif
    WG_FOR(p, P) s.pvote[p] = 0;
    WG_SYNC();
    // one fetched segment against the candidate sites: match_informative_sites (site_searcher.py:50-78) + phase_by_reads (snv_phaser.py:16-70)
    auto join_seg = [&](const SegHdr &hd, int hb, int p) {
        int qp, L, Rr;
        const int nm = uz_bsearch(hd.start, hd.end, s.cpos, nc, qp, L, Rr);
        if (nm <= 0) return;
        bool dad_alt = false, mom_alt = false;
        for (int ci = L; ci <= Rr; ci++) {
            if (s.cflag[ci] & UZ_CF_ALT_DAD) dad_alt = true; else mom_alt = true;
        }
        if (dad_alt && mom_alt) return; // site_searcher.py:74-75
        wg_atomic_add(&s.misc[1], 1);
        for (int ci = L; ci <= Rr; ci++) {
            const int rp = uz_qidx_h(R, hd, s.cpos[ci]); // snv_phaser.py:28-33
            if (rp < 0 || rp >= hd.l_seq) continue;
            const uint8_t b = uz_base(R, RowRef{hd.sq_off, hd.umask}, rp);
            bool from_ref;
            if (b == s.cref[ci]) from_ref = true;       // :41-42
            else if (b == s.calt[ci]) from_ref = false; // :43-44
            else continue;
            const bool alt_is_dad = (s.cflag[ci] & UZ_CF_ALT_DAD) != 0;
            const bool to_alt_parent = (from_ref && hb == 0) || (!from_ref && hb == 1); // :52-69
            const bool to_dad = to_alt_parent ? alt_is_dad : !alt_is_dad;
            wg_or_flag(s.pvote, p, to_dad ? 1u : 2u);
            wg_atomic_or(&s.cvote[ci], to_dad ? 1u : 2u);
        }
    };
    if (a.no_extended) {
        WG_FOR(it, nI) join_seg(uz_hdr(R, s.i_seg[it]), s.i_hb[it] & 1, (int)s.i_pair[it]);
    } else {
        // the (pair, haplotype) items of the grouped pairs, compacted first: a DNM has ~330 pairs, a few dozen of them grouped
        int nF = 0;
        WG_ROUNDS(it, 2 * P, act) {
            const int p = it >> 1;
            nF += wg_count(act && (s.grp[p] & (1u << (it & 1))) && s.pf0[p]);
        }
        ar_t<LDS>(ar, s.f_item, (size_t)nF + 1);
        if (LDS && ar.fail) return 1;
        {
            int fb = 0;
            WG_ROUNDS(it, 2 * P, act) {
                const int p = it >> 1;
                const bool on = act && (s.grp[p] & (1u << (it & 1))) && s.pf0[p];
                const int k = wg_rank(on, fb);
                if (on) s.f_item[k] = (uint32_t)it;
            }
        }
        WG_SYNC();
        // two items per lane and round: the first segments' headers, then the second segments' (the first ones' mates) beside the
        // work on the first
        constexpr int KF = 2;
        const int lane = wg_lane_opaque();
        for (int ib = 0; ib < nF; ib += KF * WG_NT) {
            int pv[KF], hbv[KF];
            bool act[KF];
            SegHdr hd0[KF], hd1[KF];
#pragma unroll
            for (int u = 0; u < KF; u++) {
                const int it = ib + u * WG_NT + lane;
                act[u] = it < nF;
                const uint32_t v = act[u] ? s.f_item[it] : 0u;
                pv[u] = (int)(v >> 1); hbv[u] = (int)(v & 1u);
                hd0[u] = uz_hdr(R, act[u] ? primary(pv[u]) : 0);
            }
#pragma unroll
            for (int u = 0; u < KF; u++) hd1[u] = uz_hdr(R, (act[u] && hd0[u].mate >= 0) ? hd0[u].mate : 0);
#pragma unroll
            for (int u = 0; u < KF; u++) {
                if (!act[u]) continue;
                join_seg(hd0[u], hbv[u], pv[u]);
                if (hd0[u].mate >= 0) join_seg(hd1[u], hbv[u], pv[u]);
            }
        }
    }
    WG_SYNC();
    UZ_TICK(15); // F.join
    const int n_match = s.misc[1];
    WG_SYNC();
    if (n_match <= 0) {
        WG_T0 a.status[d] = UZ_ST_NO_OVERLAP; // snv_phaser.py:158-166
        return 0;
    }
    // unique positions: the votes of records sharing a position count once, at the first of the run
    auto cvote_at = [&](int ci) -> uint32_t {
        if (ci > 0 && s.cpos[ci - 1] == s.cpos[ci]) return 0u;
        uint32_t v = 0;
        for (int k = ci; k < nc && s.cpos[k] == s.cpos[ci]; k++) v |= s.cvote[k];
        return v;
    };
    // counts (and optional lists): a lane owns one contiguous slice of the pairs and one of the
    // candidates, so a single six-way scan ACROSS LANES yields every list offset; pairs are in
    // ascending qname order, candidates in position order, and slices keep that order
    int cnt[6];
    {
        int plo, phi, clo, chi;
        wg_chunk(P, plo, phi);
        wg_chunk(nc, clo, chi);
        int c6[6] = {0, 0, 0, 0, 0, 0}, off[6], tot[6];
        for (int i = plo; i < phi; i++) {
            const uint32_t v = s.pvote[i], g = s.grp[i];
            c6[0] += (int)(v & 1u); c6[1] += (int)((v >> 1) & 1u);
            c6[4] += (int)(g & 1u); c6[5] += (int)((g >> 1) & 1u);
        }
        for (int i = clo; i < chi; i++) {
            const uint32_t v = cvote_at(i);
            c6[2] += (int)(v & 1u); c6[3] += (int)((v >> 1) & 1u);
        }
        wg_lane_exscan<6>(c6, off, tot, sh);
#pragma unroll
        for (int k = 0; k < 6; k++) cnt[k] = (k < 4 || a.want_lists) ? tot[k] : 0;
        if (a.want_lists) {
            // one bump allocation per DNM: the upper bound 2P + 2nc + 2P
            WG_T0 {
                const unsigned long long need = (unsigned long long)(4 * P + 2 * nc);
                const unsigned long long at = wg_atomic_add64(a.pool_cursor, need);
                a.list_start[d] = (at + need <= a.pool_cap) ? (long long)at : -1;
                for (int k = 0; k < 6; k++) a.list_len[6 * d + k] = cnt[k];
            }
            WG_SYNC();
            const long long base = a.list_start[d];
            if (base >= 0) {
                long long o[6], run = base;
#pragma unroll
                for (int k = 0; k < 6; k++) { o[k] = run + off[k]; run += tot[k]; }
                for (int i = plo; i < phi; i++) {
                    const uint32_t v = s.pvote[i], g = s.grp[i];
                    const int32_t name = (int32_t)s.pq[i];
                    if (v & 1u) a.pool[o[0]++] = name;
                    if (v & 2u) a.pool[o[1]++] = name;
                    if (g & 1u) a.pool[o[4]++] = name;
                    if (g & 2u) a.pool[o[5]++] = name;
                }
                for (int i = clo; i < chi; i++) {
                    const uint32_t v = cvote_at(i);
                    if (v & 1u) a.pool[o[2]++] = s.cpos[i];
                    if (v & 2u) a.pool[o[3]++] = s.cpos[i];
                }
            }
        }
    }
    UZ_TICK(16); // F.count
    WG_T0 {
        a.status[d] = UZ_ST_OK;
        for (int k = 0; k < 4; k++) a.counts[4 * d + k] = cnt[k];
        // summarize_record, read-backed branch: unfazed.py:206-234
        const long long dr = cnt[0], mr = cnt[1], r = a.evidence_min_ratio;
        if (dr > 0 && dr >= r * mr) { a.origin[d] = UZ_OR_DAD; a.evidence[d] = cnt[2]; }
        else if (mr > 0 && mr >= r * dr) { a.origin[d] = UZ_OR_MOM; a.evidence[d] = cnt[3]; }
        else if (dr > 0 && mr > 0) { a.origin[d] = UZ_OR_AMBIGUOUS; a.evidence[d] = (int32_t)(dr + mr); }
    }
    return 0;
}
