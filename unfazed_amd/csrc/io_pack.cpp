// io_pack.cpp -- host side of the staged record format (uz_reads_packed_view, include/uz_types.h):
//   uz_reads_pack      ASCII table -> packed columns (4-bit bases, quality-below-threshold plane, CIGAR words
//                      back to back), into buffers the caller owns (pinned memory for the upload)
//   uz_reads_select_*  the records a list of fetches can return plus their mates -- what
//                      `bamfile.fetch(chrom, lo, hi)` + `bamfile.mate(read)` hand the reference
//                      (read_collector.py:385, :167, :400, :185): nothing else of an alignment file is ever
//                      looked at, so nothing else needs to be staged.
// Host only (g++), threads over records.
#include <atomic>
#include <unordered_map>

#include "io_common.hpp"
#include "pack.hpp"

using namespace uzio;

namespace {

template <typename F>
int guarded(F &&fn) {
    try {
        fn();
        return UZ_IO_OK;
    } catch (const IoError &e) {
        last_error = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        last_error = e.what();
        return UZ_IO_E_FORMAT;
    }
}

// writable aliases of the fields of an output view
template <typename T>
T *w(const T *p) { return const_cast<T *>(p); }

// exclusive prefix sums over [0, n) of n_cigar, of the quality-plane units (every record) and of the seq4 units (records
// that carry bases: aux == nullptr means all); the arrays have n + 1 entries
void offsets(int64_t n, const uint16_t *n_cigar, const uint16_t *l_seq, const uint8_t *aux, int threads, std::vector<uint64_t> &coff,
             std::vector<uint64_t> &uoff, std::vector<uint64_t> &soff) {
    coff.assign((size_t)n + 1, 0);
    uoff.assign((size_t)n + 1, 0);
    soff.assign((size_t)n + 1, 0);
    const int wk = workers_for(n, threads, 1 << 16);
    std::vector<uint64_t> ca((size_t)wk + 1, 0), ua((size_t)wk + 1, 0), sa((size_t)wk + 1, 0);
    auto seq_units = [&](int64_t i) -> uint64_t { return (aux && (aux[i] & UZ_AUX_NO_SEQ)) ? 0 : UZ_ROW_UNITS(l_seq[i]); };
    parallel_slices(n, wk, [&](int64_t lo, int64_t hi, int k) {
        uint64_t a = 0, b = 0, c = 0;
        for (int64_t i = lo; i < hi; i++) { a += n_cigar[i]; b += UZ_ROW_UNITS(l_seq[i]); c += seq_units(i); }
        ca[(size_t)k + 1] = a; ua[(size_t)k + 1] = b; sa[(size_t)k + 1] = c;
    });
    for (int k = 0; k < wk; k++) { ca[(size_t)k + 1] += ca[k]; ua[(size_t)k + 1] += ua[k]; sa[(size_t)k + 1] += sa[k]; }
    parallel_slices(n, wk, [&](int64_t lo, int64_t hi, int k) {
        uint64_t a = ca[k], b = ua[k], c = sa[k];
        for (int64_t i = lo; i < hi; i++) {
            coff[i] = a; uoff[i] = b; soff[i] = c;
            a += n_cigar[i]; b += UZ_ROW_UNITS(l_seq[i]); c += seq_units(i);
        }
    });
    coff[n] = ca[wk]; uoff[n] = ua[wk]; soff[n] = sa[wk];
}

} // namespace

// a packed table opened as the source of selections: the CIGAR / row offsets of its records, computed once
struct uz_psrc {
    uz_reads_packed_view v;
    std::vector<uint32_t> coff, uoff, soff; // [n + 1]: CIGAR words, quality-plane units, seq4 units
    std::vector<uint64_t> loff;             // [n + 1] list-form sources: first entry of every record in qlow_pos (listed records own n_low entries)
};
// low-quality bases of record i of a source table, and whether its positions are / can be listed
static inline int src_low_count(const uz_psrc *src, int64_t i) {
    const uz_reads_packed_view *f = &src->v;
    if (f->n_low) return f->n_low[i];
    int c = 0;
    const uint8_t *row = f->qlow + (size_t)src->uoff[i] * UZ_QLOW_UNIT_BYTES;
    const int ls = f->l_seq[i];
    for (int k = 0; k < ls; k++) c += (row[k >> 3] >> (k & 7)) & 1;
    return c > 255 ? 255 : c;
}

// the low-quality positions of record i (ascending), each handed to fn -- from the source's lists or its plane
template <typename F>
static inline void src_low_positions(const uz_psrc *src, int64_t i, F &&fn) {
    const uz_reads_packed_view *f = &src->v;
    if (f->n_low) {
        if ((f->aux[i] & UZ_AUX_NO_SEQ) || f->n_low[i] > UZ_QLOW_LIST_MAX) return; // (no list)
        for (int e = 0; e < (int)f->n_low[i]; e++) {
            const uint64_t from = src->loff[(size_t)i] + (uint64_t)e;
            fn(f->qlow_pos_wide ? (int)f->qlow_pos[2 * from] | ((int)f->qlow_pos[2 * from + 1] << 8) : (int)f->qlow_pos[from]);
        }
        return;
    }
    const uint8_t *row = f->qlow + (size_t)src->uoff[i] * UZ_QLOW_UNIT_BYTES;
    const int ls = f->l_seq[i];
    for (int b = 0; b < ls; b++)
        if ((row[b >> 3] >> (b & 7)) & 1) fn(b);
}
// is base b of a record inside the 32-base units the mask keeps?
static inline bool unit_kept(uint16_t m16, int b) { return m16 == UZ_UMASK_ALL || ((m16 >> (b >> 5)) & 1); }

struct uz_select {
    const uz_psrc *src = nullptr;
    int64_t n_sel = 0;
    uint64_t n_cigar = 0, n_units = 0, n_seq = 0;
    std::vector<int32_t> index; // kept records, ascending (indices into the source table)
    std::vector<uint8_t> bases; // per kept record: 1 = its bases are staged (a fetch returns it), 0 = reachable only as a mate
    std::vector<uint16_t> umask;        // per kept record: staged 32-base units of its rows (UZ_UMASK_ALL: every unit; empty vector: no masks asked for)
    // dictionary form of the small columns (uz_reads_packed_view.tup): built by the plan when asked for
    int tuples = 0;                     // 0 none; else bit 0 set, bit 1: aux carries the simple-CIGAR code, bit 2: n_low is part of the combination
    std::vector<uint16_t> tup_idx;      // per kept record
    std::vector<uint64_t> tup_key;      // per combination: flag | l_seq << 16 | n_cigar << 32 | mapq << 48 | aux << 56
    std::vector<uint8_t> tup_low;
    std::vector<uint16_t> tup_um;       // (unit mask of the combination when the selection has masks)
    int64_t n_esc16 = 0;                // escapes of the 16-bit difference form of start / tlen / mate / qname (uz_d16_of)
    int64_t n_esc16_start8 = 0;         // ... when the start differences travel in eight bits (start_d8)
    int64_t n_esc16_narrow8 = 0;        // ... and the mate / name-id differences too (mate_d8, qname_d8)
    // the pair form (pair_d8): a new name's id is the number of new names before it, so the output numbers the names of the selection
    // by first appearance; that keeps the ORDER of the source's ids (what the read stage sorts pairs by) only when the source's ids
    // ascend by first appearance too -- checked by the plan
    int pair8_ok = 0;
    int64_t n_esc16_pair8 = 0;
    std::vector<uint8_t> is_new;        // per kept record: the first of the selection with its name
    std::vector<uint32_t> new_ids;      // source id of every new name, ascending = in order of appearance: output id -> source id
    int64_t n_cigar_simple = 0;         // kept records whose CIGAR is one M / = / X over the read (their words can stay home)
    int end_derivable = 1;              // every kept record's end is what its CIGAR gives (the output may leave the column out)
    std::vector<uint8_t> n_low;         // per kept record: low-quality bases (saturated), for the list form of the output
    int64_t n_qpos = 0;                 // entries of the output's qlow_pos
    std::vector<int64_t> exc_lo, exc_n; // seq2 sources: per kept record, its slice of the source's exception list (0 entries without bases)
    int64_t n_exc = 0;
    // the list form of the bases (uz_types.h bl_*; planned with unit_masks & 4 on a two-bit source): per kept record the number of listed
    // bases (0: its staged units travel as rows), the query indices back to back, their first entry per record
    std::vector<uint8_t> bl_n;
    std::vector<uint16_t> bl_pos;
    std::vector<int64_t> bl_off;        // [n_sel + 1]
    std::vector<uint8_t> tup_nb;        // (list count of the combination)
    int64_t n_bl_units = 0;             // row units of the listed records (not in n_seq)
    int bl_wide = 0;
};

// start / tlen / mate / qname of kept record k as 16-bit differences (uz_reads_packed_view.start_d ...): v[c] the column values,
// e[c] the escape value where v[c] == UZ_D16_ESC (columns 0 start, 1 tlen, 2 mate, 3 qname); returns the number of escapes
// start8: the start difference is to fit EIGHT bits (0 .. 254; uz_reads_packed_view.start_d8): anything else escapes
// narrow8: mate and name-id differences are to fit a signed byte (-126 .. 127; UZ_D8S_ESC / UZ_D8S_NONE); v[] then holds those codes
static inline int uz_d16_of(const uz_select *s, int64_t k, int16_t v[4], int32_t e[4], bool start8 = false, bool narrow8 = false) {
    const uz_reads_packed_view *f = &s->src->v;
    const int64_t i = s->index[(size_t)k];
    const int64_t ip = k > 0 ? s->index[(size_t)k - 1] : -1;
    int n = 0;
    auto put = [&](int c, int64_t d, int64_t esc_val) {
        if (narrow8 && c >= 2) {
            if (d > -127 && d <= 127) v[c] = (int16_t)d;
            else { v[c] = (int16_t)UZ_D8S_ESC; e[c] = (int32_t)esc_val; n++; }
            return;
        }
        if (d > -32767 && d <= 32767) v[c] = (int16_t)d;
        else { v[c] = (int16_t)UZ_D16_ESC; e[c] = (int32_t)esc_val; n++; }
    };
    const int64_t ds = (int64_t)f->start[i] - (ip >= 0 ? (int64_t)f->start[ip] : 0);
    if (!start8) put(0, ds, ds);
    else if (ds >= 0 && ds <= 254) v[0] = (int16_t)ds;
    else { v[0] = (int16_t)UZ_D16_ESC; e[0] = (int32_t)ds; n++; }
    put(1, f->tlen[i], f->tlen[i]);
    const int32_t mt = f->mate[i];
    int64_t nm = -1;
    if (mt >= 0 && mt < f->n_segs) { // new index of the mate: its rank in the (sorted) selection
        auto it = std::lower_bound(s->index.begin(), s->index.end(), mt);
        if (it != s->index.end() && *it == mt) nm = it - s->index.begin();
    }
    if (nm < 0) v[2] = narrow8 ? (int16_t)UZ_D8S_NONE : (int16_t)UZ_D16_NONE;
    else put(2, nm - k, nm);
    const int32_t dq = (int32_t)(f->qname[i] - (ip >= 0 ? f->qname[ip] : 0u)); // modulo 2^32
    put(3, dq, dq);
    return n;
}

// the pair form's view of kept record k (io_common.hpp)
static inline int64_t sel_rank(const uz_select *s, int64_t src_index) {
    auto it = std::lower_bound(s->index.begin(), s->index.end(), src_index, [](int32_t v, int64_t key) { return (int64_t)v < key; });
    return (it != s->index.end() && *it == src_index) ? (int64_t)(it - s->index.begin()) : -1;
}
static inline PairRec sel_pair_rec(const uz_select *s, int64_t k) {
    const uz_reads_packed_view *f = &s->src->v;
    const int64_t i = s->index[(size_t)k];
    PairRec r;
    const int32_t mt = f->mate[i];
    r.mate = (mt >= 0 && mt < f->n_segs) ? sel_rank(s, mt) : -1;
    r.start = f->start[i]; r.end = f->end[i]; r.tlen = f->tlen[i];
    r.is_new = s->is_new[(size_t)k] != 0;
    r.qid = (uint32_t)(std::lower_bound(s->new_ids.begin(), s->new_ids.end(), f->qname[i]) - s->new_ids.begin()); // rank of the source id = output id
    return r;
}
static inline int sel_link_of(const uz_select *s, int64_t k, uint8_t &start8, uint8_t &pair, int32_t e[4], bool has[4]) {
    const uz_reads_packed_view *f = &s->src->v;
    const int64_t i = s->index[(size_t)k];
    const int64_t ip = k > 0 ? s->index[(size_t)k - 1] : -1;
    const int64_t ds = (int64_t)f->start[i] - (ip >= 0 ? (int64_t)f->start[ip] : 0);
    int n = 0;
    has[0] = !(ds >= 0 && ds <= 254);
    if (has[0]) { start8 = (uint8_t)UZ_D8_ESC; e[0] = (int32_t)ds; n++; }
    else start8 = (uint8_t)ds;
    pair = pair8_code(k, [&](int64_t j) { return sel_pair_rec(s, j); });
    return n + pair8_escapes(pair, sel_pair_rec(s, k), e, has);
}

extern "C" {

int uz_reads_pack_sizes(const uz_reads_view *in, int64_t *n_cigar_total, int64_t *n_row_units) {
    return guarded([&] {
        if (!in || !n_cigar_total || !n_row_units) fail(UZ_IO_E_ARG, "null argument");
        uint64_t a = 0, b = 0;
        for (int64_t i = 0; i < in->n_segs; i++) { a += in->n_cigar[i]; b += UZ_ROW_UNITS(in->l_seq[i]); }
        *n_cigar_total = (int64_t)a; *n_row_units = (int64_t)b;
    });
}

int uz_reads_pack_lists(const uz_reads_view *in, int min_base_qual, int threads, int64_t *n_qlow_pos, int32_t *wide) {
    return guarded([&] {
        if (!in || !n_qlow_pos || !wide) fail(UZ_IO_E_ARG, "null argument");
        threads = resolve_threads(threads);
        const int64_t n = in->n_segs;
        const int thr = min_base_qual < 0 ? 0 : (min_base_qual > 255 ? 256 : min_base_qual);
        const int wk = workers_for(n, threads, 4096);
        std::vector<int64_t> part((size_t)wk + 1, 0), wd((size_t)wk + 1, 0);
        parallel_slices(n, wk, [&](int64_t lo, int64_t hi, int k) {
            int64_t c = 0, w2 = 0;
            for (int64_t i = lo; i < hi; i++) {
                const uint8_t *q = in->qual + ((size_t)in->sq_off16[i] << 4);
                int low = 0;
                for (int b = 0; b < (int)in->l_seq[i]; b++) low += (int)q[b] < thr;
                if (low <= UZ_QLOW_LIST_MAX) c += low;
                if (in->l_seq[i] > 256) w2 = 1;
            }
            part[(size_t)k] = c; wd[(size_t)k] = w2;
        });
        int64_t tot = 0, w2 = 0;
        for (int k = 0; k < wk; k++) { tot += part[(size_t)k]; w2 |= wd[(size_t)k]; }
        *n_qlow_pos = tot; *wide = (int32_t)w2;
    });
}

int uz_reads_pack_cigar_omitted(const uz_reads_view *in, int threads, int64_t *n_omitted) {
    return guarded([&] {
        if (!in || !n_omitted) fail(UZ_IO_E_ARG, "null argument");
        threads = resolve_threads(threads);
        const int wk = workers_for(in->n_segs, threads, 1 << 14);
        std::vector<int64_t> part((size_t)wk + 1, 0);
        parallel_slices(in->n_segs, wk, [&](int64_t lo, int64_t hi, int k) {
            int64_t c = 0;
            for (int64_t i = lo; i < hi; i++)
                c += uz_cigar_simple_code(in->n_cigar[i], in->n_cigar[i] ? in->cigar[in->cigar_off[i]] : 0u, in->l_seq[i]) != 0;
            part[(size_t)k] = c;
        });
        int64_t tot = 0;
        for (int k = 0; k < wk; k++) tot += part[(size_t)k];
        *n_omitted = tot;
    });
}

int uz_reads_pack_end_derivable(const uz_reads_view *in, int threads, int32_t *yes) {
    return guarded([&] {
        if (!in || !yes) fail(UZ_IO_E_ARG, "null argument");
        threads = resolve_threads(threads);
        std::atomic<int> bad{0};
        parallel_slices(in->n_segs, workers_for(in->n_segs, threads, 1 << 14), [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; i++)
                if (in->end[i] != uz_bam_endpos(in->start[i], in->flag[i], in->n_cigar[i], in->cigar + in->cigar_off[i])) { bad.store(1); return; }
        });
        *yes = bad.load() ? 0 : 1;
    });
}

int uz_reads_pack_exceptions(const uz_reads_view *in, int threads, int64_t *n_exc) {
    return guarded([&] {
        if (!in || !n_exc) fail(UZ_IO_E_ARG, "null argument");
        threads = resolve_threads(threads);
        const int64_t n = in->n_segs;
        const int wk = workers_for(n, threads, 4096);
        std::vector<int64_t> part((size_t)wk + 1, 0);
        parallel_slices(n, wk, [&](int64_t lo, int64_t hi, int k) {
            int64_t c = 0;
            for (int64_t i = lo; i < hi; i++) {
                const uint8_t *sq = in->seq + ((size_t)in->sq_off16[i] << 4);
                for (int b = 0; b < (int)in->l_seq[i]; b++) c += uz_ascii_seq2(sq[b]) == 0xFF;
            }
            part[(size_t)k] = c;
        });
        int64_t tot = 0;
        for (int k = 0; k < wk; k++) tot += part[(size_t)k];
        *n_exc = tot;
    });
}

int uz_reads_pack(const uz_reads_view *in, int min_base_qual, int threads, uz_reads_packed_view *out) {
    return guarded([&] {
        if (!in || !out) fail(UZ_IO_E_ARG, "null argument");
        const int64_t n = in->n_segs;
        threads = resolve_threads(threads);
        std::vector<uint64_t> coff, uoff, soff;
        offsets(n, in->n_cigar, in->l_seq, nullptr, threads, coff, uoff, soff);
        if (coff[n] >= ((uint64_t)1 << 32) || uoff[n] >= ((uint64_t)1 << 32)) fail(UZ_IO_E_RANGE, "table exceeds the 32-bit CIGAR / row offsets");
        std::vector<uint8_t> scode;
        if (out->cigar_compact) { // words of the simple records stay home: offsets over the others
            scode.assign((size_t)n + 1, 0);
            uint64_t at = 0, om = 0;
            for (int64_t i = 0; i < n; i++) {
                const uint32_t code = uz_cigar_simple_code(in->n_cigar[i], in->n_cigar[i] ? in->cigar[in->cigar_off[i]] : 0u, in->l_seq[i]);
                scode[(size_t)i] = (uint8_t)code;
                coff[(size_t)i] = at;
                if (code) om++; else at += in->n_cigar[i];
            }
            coff[(size_t)n] = at;
            if ((int64_t)om != out->n_cigar_omitted) fail(UZ_IO_E_ARG, "output view sized for %lld omitted CIGAR words, the table has %llu (uz_reads_pack_cigar_omitted)", (long long)out->n_cigar_omitted, (unsigned long long)om);
        } else
            out->n_cigar_omitted = 0;
        if ((int64_t)coff[n] != out->n_cigar_total || (int64_t)uoff[n] != out->n_row_units)
            fail(UZ_IO_E_ARG, "output view sized for %lld / %lld CIGAR words / row units, the table has %llu / %llu", (long long)out->n_cigar_total,
                 (long long)out->n_row_units, (unsigned long long)coff[n], (unsigned long long)uoff[n]);
        out->n_segs = n; out->n_contigs = in->n_contigs; out->min_base_qual = min_base_qual; out->n_qnames = in->n_qnames;
        out->n_seq_units = out->n_row_units; // the ASCII form carries every record's bases
        memcpy(w(out->contig_off), in->contig_off, ((size_t)in->n_contigs + 1) * sizeof(int64_t));
        if (in->n_contigs) memcpy(w(out->max_span), in->max_span, (size_t)in->n_contigs * sizeof(int32_t));
        const int thr = min_base_qual < 0 ? 0 : (min_base_qual > 255 ? 256 : min_base_qual);
        std::atomic<int> bad{0};
        const bool two_bit = out->seq2 != nullptr;
        const bool lists = out->n_low != nullptr;
        std::vector<uint64_t> loff;
        if (lists) { // counts first, then the offsets of the lists
            parallel_slices(n, workers_for(n, threads, 4096), [&](int64_t lo, int64_t hi, int) {
                for (int64_t i = lo; i < hi; i++) {
                    const uint8_t *q = in->qual + ((size_t)in->sq_off16[i] << 4);
                    int low = 0;
                    for (int b = 0; b < (int)in->l_seq[i]; b++) low += (int)q[b] < thr;
                    w(out->n_low)[i] = (uint8_t)(low > 255 ? 255 : low);
                    if (in->l_seq[i] > 256 && !out->qlow_pos_wide) bad.store(2);
                }
            });
            if (bad.load() == 2) fail(UZ_IO_E_ARG, "reads longer than 256 bases need qlow_pos_wide (uz_reads_pack_lists)");
            loff.assign((size_t)n + 1, 0);
            for (int64_t i = 0; i < n; i++) loff[(size_t)i + 1] = loff[(size_t)i] + (out->n_low[i] <= UZ_QLOW_LIST_MAX ? out->n_low[i] : 0);
            if ((int64_t)loff[(size_t)n] != out->n_qlow_pos)
                fail(UZ_IO_E_ARG, "output view sized for %lld listed positions, the table has %llu (uz_reads_pack_lists)", (long long)out->n_qlow_pos,
                     (unsigned long long)loff[(size_t)n]);
        }
        const int wk_pack = workers_for(n, threads, 4096);
        // two-bit rows: the bases that are not A/C/G/T, per slice (slices are contiguous record ranges, so their lists joined in
        // slice order are sorted by record)
        struct Exc { uint32_t rec; uint16_t pos; uint8_t code; };
        std::vector<std::vector<Exc>> exc((size_t)wk_pack);
        parallel_slices(n, wk_pack, [&](int64_t lo, int64_t hi, int slice) {
            for (int64_t i = lo; i < hi; i++) {
                w(out->start)[i] = in->start[i]; w(out->tlen)[i] = in->tlen[i];
                if (out->end) w(out->end)[i] = in->end[i];
                else if (in->end[i] != uz_bam_endpos(in->start[i], in->flag[i], in->n_cigar[i], in->cigar + in->cigar_off[i])) bad.store(3);
                w(out->mate)[i] = in->mate[i]; w(out->qname)[i] = in->qname[i]; w(out->flag)[i] = in->flag[i];
                w(out->l_seq)[i] = in->l_seq[i]; w(out->n_cigar)[i] = in->n_cigar[i]; w(out->mapq)[i] = in->mapq[i];
                w(out->aux)[i] = in->aux[i];
                if (!scode.empty() && scode[(size_t)i]) w(out->aux)[i] = (uint8_t)(in->aux[i] | (scode[(size_t)i] << UZ_AUX_SIMPLE_SHIFT));
                else
                    for (int k = 0; k < (int)in->n_cigar[i]; k++) w(out->cigar)[coff[i] + k] = in->cigar[(size_t)in->cigar_off[i] + k];
                const size_t row = (size_t)in->sq_off16[i] << 4;
                int rc;
                if (two_bit)
                    rc = uz_pack_rows_host2(in->seq + row, in->qual + row, in->l_seq[i], thr, w(out->seq2) + uoff[i] * UZ_SEQ2_UNIT_BYTES,
                                            lists ? (uint8_t *)nullptr : w(out->qlow) + uoff[i] * UZ_QLOW_UNIT_BYTES,
                                            [&](int pos, uint8_t code) { exc[(size_t)slice].push_back(Exc{(uint32_t)i, (uint16_t)pos, code}); });
                else
                    rc = uz_pack_rows_host(in->seq + row, in->qual + row, in->l_seq[i], thr, w(out->seq4) + uoff[i] * UZ_SEQ4_UNIT_BYTES,
                                           lists ? (uint8_t *)nullptr : w(out->qlow) + uoff[i] * UZ_QLOW_UNIT_BYTES);
                if (rc != 0) bad.store(1);
                if (lists && out->n_low[i] <= UZ_QLOW_LIST_MAX) {
                    uint64_t at = loff[(size_t)i];
                    const uint8_t *q = in->qual + row;
                    for (int b = 0; b < (int)in->l_seq[i]; b++)
                        if ((int)q[b] < thr) {
                            if (out->qlow_pos_wide) { w(out->qlow_pos)[2 * at] = (uint8_t)(b & 255); w(out->qlow_pos)[2 * at + 1] = (uint8_t)(b >> 8); }
                            else w(out->qlow_pos)[at] = (uint8_t)b;
                            at++;
                        }
                }
            }
        });
        if (bad.load() == 3) fail(UZ_IO_E_ARG, "the `end` column was left out but a record's end is not what its CIGAR gives (uz_reads_pack_end_derivable)");
        if (bad.load()) fail(UZ_IO_E_RANGE, "SEQ holds a character outside BAM's 16-code alphabet");
        if (two_bit) {
            int64_t tot = 0;
            for (auto &e : exc) tot += (int64_t)e.size();
            if (tot != out->n_exc) fail(UZ_IO_E_ARG, "output view sized for %lld listed bases, the table has %lld (uz_reads_pack_exceptions)", (long long)out->n_exc, (long long)tot);
            int64_t at = 0;
            for (auto &e : exc)
                for (const Exc &x : e) { w(out->exc_rec)[at] = x.rec; w(out->exc_pos)[at] = x.pos; w(out->exc_code)[at] = x.code; at++; }
        } else
            out->n_exc = 0;
    });
}

int uz_reads_source_open(const uz_reads_packed_view *full, int threads, uz_psrc **out) {
    return guarded([&] {
        if (!full || !out) fail(UZ_IO_E_ARG, "null argument");
        threads = resolve_threads(threads);
        auto src = new uz_psrc();
        src->v = *full;
        std::vector<uint64_t> coff, uoff, soff;
        offsets(full->n_segs, full->n_cigar, full->l_seq, full->aux, threads, coff, uoff, soff);
        if (coff.back() >= ((uint64_t)1 << 32) || uoff.back() >= ((uint64_t)1 << 32)) { delete src; fail(UZ_IO_E_RANGE, "table exceeds the 32-bit CIGAR / row offsets"); }
        if ((int64_t)soff.back() != full->n_seq_units) { delete src; fail(UZ_IO_E_ARG, "n_seq_units does not match the aux column"); }
        if (!full->end) { delete src; fail(UZ_IO_E_ARG, "a source of selections needs the `end` column"); }
        if (full->cigar_compact) { delete src; fail(UZ_IO_E_ARG, "a source of selections needs every CIGAR word (cigar_compact = 0)"); }
        if (!full->qlow && !full->n_low) { delete src; fail(UZ_IO_E_ARG, "the table has neither the quality plane nor its list form"); }
        if (full->n_low) {
            src->loff.assign((size_t)full->n_segs + 1, 0);
            for (int64_t i = 0; i < full->n_segs; i++) {
                const bool listed = !(full->aux[i] & UZ_AUX_NO_SEQ) && full->n_low[i] <= UZ_QLOW_LIST_MAX;
                src->loff[(size_t)i + 1] = src->loff[(size_t)i] + (listed ? full->n_low[i] : 0);
            }
            if ((int64_t)src->loff.back() != full->n_qlow_pos) { delete src; fail(UZ_IO_E_ARG, "n_qlow_pos does not match the n_low / aux columns"); }
        }
        if (full->seq2 && full->n_exc > 0)
            for (int64_t e = 1; e < full->n_exc; e++)
                if (full->exc_rec[e] < full->exc_rec[e - 1]) { delete src; fail(UZ_IO_E_ARG, "exc_rec is not ascending"); }
        src->coff.assign(coff.begin(), coff.end());
        src->uoff.assign(uoff.begin(), uoff.end());
        src->soff.assign(soff.begin(), soff.end());
        *out = src;
    });
}
void uz_reads_source_close(uz_psrc *s) { delete s; }

int uz_reads_select_plan(const uz_psrc *src, int64_t n_fetch, const int32_t *contig, const int32_t *lo, const int32_t *hi, int all_bases,
                         int unit_masks, const uint16_t *extra, int tuples, int threads, uz_select **out) {
    return guarded([&] {
        if (!src || !out || (n_fetch > 0 && (!contig || !lo || !hi))) fail(UZ_IO_E_ARG, "null argument");
        threads = resolve_threads(threads);
        const uz_reads_packed_view *full = &src->v;
        const int64_t n = full->n_segs;
        std::vector<uint8_t> keep((size_t)n + 1, 0);
        uint8_t *kp = keep.data();
        // unit masks: the read stage reads the bases of a record at the fetched position (and `extra` bases on: the alleles of a
        // DNM) only -- position hi - 1 of a one- or two-base fetch.  For a record whose single CIGAR operation spans the read the
        // query index is position - start; every other record (and every record of a wider fetch: SV breakpoints) keeps all units.
        const bool masks = (unit_masks & 1) && !all_bases;
        // unit_masks & 2: a fetch wider than two bases (the +-cutoff fetches around an SV's breakpoints) stages NO unit of the records
        // it returns -- collect_reads_sv looks at flags, CIGARs and mates only (read_collector.py:476-596); the bases of such a record
        // are read at the het sites it overlaps, which have one-base fetches of their own
        const bool wide_none = (unit_masks & 2) != 0;
        std::vector<uint16_t> um;
        if (masks) um.assign((size_t)n + 1, 0);
        uint16_t *ump = um.data();
        std::atomic<int64_t> rmin{n}, rmax{0}; // record range the marks fall into
        auto widen = [&](int64_t a, int64_t b) {
            int64_t cur = rmin.load();
            while (a < cur && !rmin.compare_exchange_weak(cur, a)) {}
            cur = rmax.load();
            while (b > cur && !rmax.compare_exchange_weak(cur, b)) {}
        };
        // pysam fetch(contig, lo, hi): records with start < hi and end > lo; their starts lie in [lo - max_span, hi)
        parallel_slices(n_fetch, workers_for(n_fetch, threads, 256), [&](int64_t f0, int64_t f1, int) {
            int64_t a = n, b = 0;
            for (int64_t f = f0; f < f1; f++) {
                const int tid = contig[f];
                if (tid < 0 || tid >= full->n_contigs) continue;
                const int64_t c0 = full->contig_off[tid], c1 = full->contig_off[tid + 1];
                const int64_t from = (int64_t)lo[f] - full->max_span[tid];
                const int32_t *p = std::lower_bound(full->start + c0, full->start + c1, from,
                                                    [](int32_t v, int64_t key) { return (int64_t)v < key; });
                for (int64_t i = p - full->start; i < c1 && full->start[i] < hi[f]; i++)
                    if (full->end[i] > lo[f]) {
                        __atomic_store_n(&kp[i], (uint8_t)2, __ATOMIC_RELAXED); // 2: returned by a fetch, 1: reachable only as a mate
                        a = std::min(a, i); b = std::max(b, i + 1);
                        if (masks) {
                            uint16_t bits = (uint16_t)UZ_UMASK_ALL;
                            const int ls = full->l_seq[i];
                            const uint32_t cw = full->n_cigar[i] == 1 ? full->cigar[src->coff[(size_t)i]] : 0u;
                            const uint32_t op = cw & 15u;
                            const bool simple = full->n_cigar[i] == 1 && (op == 0 || op == 7 || op == 8) && (int)(cw >> 4) == ls && ls > 1 &&
                                                full->end[i] - full->start[i] == ls;
                            if (hi[f] - lo[f] > 2 && wide_none) bits = 0;
                            else if (simple && ls <= 480 && hi[f] - lo[f] <= 2) {
                                const int64_t q0 = (int64_t)hi[f] - 1 - full->start[i];
                                bits = 0;
                                if (q0 >= 0 && q0 < ls) {
                                    const int64_t q1 = std::min<int64_t>(q0 + (extra ? extra[f] : 0), ls - 1);
                                    for (int64_t u = q0 >> 5; u <= (q1 >> 5); u++) bits |= (uint16_t)(1u << u);
                                }
                            }
                            __atomic_fetch_or(&ump[i], bits, __ATOMIC_RELAXED);
                        }
                    }
            }
            if (a < b) widen(a, b);
        });
        // mates, and the mates of those (mate() is not an involution when secondary / supplementary records
        // share a name): closed under `mate` after a few rounds
        for (int round = 0; round < 8; round++) {
            std::atomic<int64_t> added{0};
            const int64_t r0 = rmin.load(), r1 = rmax.load();
            if (r0 >= r1) break;
            parallel_slices(r1 - r0, workers_for(r1 - r0, threads, 1 << 16), [&](int64_t a, int64_t b, int) {
                int64_t mine = 0, lo2 = n, hi2 = 0;
                for (int64_t i = r0 + a; i < r0 + b; i++) {
                    if (!__atomic_load_n(&kp[i], __ATOMIC_RELAXED)) continue;
                    const int32_t m = full->mate[i];
                    if (m >= 0 && m < n && !__atomic_load_n(&kp[m], __ATOMIC_RELAXED)) {
                        __atomic_store_n(&kp[m], (uint8_t)1, __ATOMIC_RELAXED);
                        mine++;
                        lo2 = std::min<int64_t>(lo2, m); hi2 = std::max<int64_t>(hi2, (int64_t)m + 1);
                    }
                }
                if (lo2 < hi2) widen(lo2, hi2);
                added += mine;
            });
            if (added.load() == 0) break;
        }
        auto sel = new uz_select();
        sel->src = src;
        for (int64_t i = rmin.load(); i < rmax.load(); i++)
            if (kp[i]) {
                // bases travel with a record a fetch returns (every base the read stage reads lies at a fetch point the
                // record overlaps: DNM position, het site, candidate = het site); with --no-extended the join reads the
                // mates of the DNM reads at candidate sites nobody fetched, so the caller asks for all of them
                const bool bases = (kp[i] == 2 || all_bases) && !(full->aux[i] & UZ_AUX_NO_SEQ);
                sel->index.push_back((int32_t)i);
                sel->bases.push_back(bases ? 1 : 0);
                sel->n_cigar += full->n_cigar[i]; sel->n_units += UZ_ROW_UNITS(full->l_seq[i]);
                uint16_t m16 = (uint16_t)UZ_UMASK_ALL;
                if (masks) { // (a record with bases that no fetched position falls into keeps none of its units: a zero mask)
                    m16 = bases ? ump[i] : (uint16_t)0;
                    if (m16 != UZ_UMASK_ALL && (uint32_t)__builtin_popcount(m16) == UZ_ROW_UNITS(full->l_seq[i])) m16 = (uint16_t)UZ_UMASK_ALL;
                    sel->umask.push_back(m16);
                }
                if (bases) sel->n_seq += m16 == UZ_UMASK_ALL ? UZ_ROW_UNITS(full->l_seq[i]) : (uint32_t)__builtin_popcount(m16);
            }
        sel->n_sel = (int64_t)sel->index.size();
        for (int64_t k = 0; k < sel->n_sel; k++) {
            const int64_t i = sel->index[(size_t)k];
            sel->n_cigar_simple += uz_cigar_simple_code(full->n_cigar[i], full->n_cigar[i] ? full->cigar[src->coff[(size_t)i]] : 0u, full->l_seq[i]) != 0;
        }
        sel->n_low.assign((size_t)sel->n_sel, 0);
        {
            std::vector<int64_t> part;
            const int wk = workers_for(sel->n_sel, threads, 4096);
            part.assign((size_t)wk + 1, 0);
            parallel_slices(sel->n_sel, wk, [&](int64_t a, int64_t b, int slice) {
                int64_t c = 0;
                for (int64_t k = a; k < b; k++) {
                    const int64_t i = sel->index[(size_t)k];
                    if (full->end[i] != uz_bam_endpos(full->start[i], full->flag[i], full->n_cigar[i], full->cigar + src->coff[(size_t)i]))
                        __atomic_store_n(&sel->end_derivable, 0, __ATOMIC_RELAXED);
                    int low = src_low_count(src, sel->index[(size_t)k]);
                    if (sel->bases[(size_t)k] && low <= UZ_QLOW_LIST_MAX) {
                        // a listed record: with unit masks only the positions inside the staged units travel -- no bit of any
                        // other unit can be asked for (uz_types.h) -- and the count that travels is the length of that list;
                        // "at most UZ_QLOW_LIST_MAX low-quality bases", all the read filter wants of the count, stays true
                        if (!sel->umask.empty() && sel->umask[(size_t)k] != UZ_UMASK_ALL) {
                            const uint16_t m16 = sel->umask[(size_t)k];
                            int kept = 0;
                            src_low_positions(src, i, [&](int b) { kept += unit_kept(m16, b); });
                            low = kept;
                        }
                        c += low;
                    }
                    sel->n_low[(size_t)k] = (uint8_t)low;
                }
                part[(size_t)slice] = c;
            });
            for (int k = 0; k < wk; k++) sel->n_qpos += part[(size_t)k];
        }
        if (full->seq2) { // the listed bases of the kept records that keep their bases
            sel->exc_lo.assign((size_t)sel->n_sel, 0);
            sel->exc_n.assign((size_t)sel->n_sel, 0);
            const uint32_t *er = full->exc_rec;
            for (int64_t k = 0; k < sel->n_sel && full->n_exc > 0; k++) {
                if (!sel->bases[(size_t)k]) continue;
                const uint32_t i = (uint32_t)sel->index[(size_t)k];
                const uint32_t *a = std::lower_bound(er, er + full->n_exc, i), *b = std::upper_bound(a, er + full->n_exc, i);
                sel->exc_lo[(size_t)k] = a - er; sel->exc_n[(size_t)k] = b - a; // (entries in units that stay home travel along: the device skips them)
                sel->n_exc += b - a;
            }
        }
        if (masks && (unit_masks & 4) && full->seq2) {
            // The bases as lists: a kept record with an explicit unit mask was returned by one- / two-base fetches only and its single CIGAR
            // operation spans the read, so every base the read stage can ask of it is (fetched position - start) .. + extra.  Those
            // positions travel instead of the units they lie in when that is fewer bytes and none of them is '=' (uz_types.h).
            std::vector<int64_t> ford((size_t)n_fetch);
            for (int64_t f = 0; f < n_fetch; f++) ford[(size_t)f] = f;
            std::sort(ford.begin(), ford.end(), [&](int64_t a, int64_t b) { return contig[a] != contig[b] ? contig[a] < contig[b] : (lo[a] != lo[b] ? lo[a] < lo[b] : a < b); });
            std::vector<int64_t> cfirst((size_t)full->n_contigs + 1, n_fetch); // first sorted fetch of every contig
            for (int64_t j = n_fetch - 1; j >= 0; j--) { const int c = contig[ford[(size_t)j]]; if (c >= 0 && c < full->n_contigs) cfirst[(size_t)c] = j; }
            sel->bl_n.assign((size_t)sel->n_sel, 0);
            sel->bl_off.assign((size_t)sel->n_sel + 1, 0);
            std::vector<uint16_t> v;
            for (int64_t k = 0; k < sel->n_sel; k++) {
                sel->bl_off[(size_t)k] = (int64_t)sel->bl_pos.size();
                const uint16_t m16 = sel->umask[(size_t)k];
                if (!sel->bases[(size_t)k] || m16 == UZ_UMASK_ALL || m16 == 0) continue;
                const int64_t i = sel->index[(size_t)k];
                const int tid = (int)(std::upper_bound(full->contig_off, full->contig_off + full->n_contigs + 1, i) - full->contig_off) - 1;
                const int32_t st = full->start[i], en = full->end[i], ls = full->l_seq[i];
                v.clear();
                int64_t j = cfirst[(size_t)tid];
                { // first fetch of the contig with lo >= st - 1 (a one- or two-base fetch that overlaps the record starts there or later)
                    int64_t a = j, b = n_fetch;
                    while (a < b) { const int64_t mid = (a + b) >> 1; const int64_t f = ford[(size_t)mid]; if (contig[f] < tid || (contig[f] == tid && lo[f] < st - 1)) a = mid + 1; else b = mid; }
                    j = a;
                }
                for (; j < n_fetch; j++) {
                    const int64_t f = ford[(size_t)j];
                    if (contig[f] != tid || lo[f] >= en) break;
                    if (hi[f] - lo[f] > 2 || hi[f] <= st) continue; // (a wider fetch contributes no unit here: wide_none -- or the mask would be "all")
                    const int64_t q0 = (int64_t)hi[f] - 1 - st;
                    if (q0 < 0 || q0 >= ls) continue;
                    const int64_t q1 = std::min<int64_t>(q0 + (extra ? extra[f] : 0), ls - 1);
                    for (int64_t q = q0; q <= q1; q++) v.push_back((uint16_t)q);
                }
                std::sort(v.begin(), v.end());
                v.erase(std::unique(v.begin(), v.end()), v.end());
                uint16_t units = 0;
                for (uint16_t q : v) units |= (uint16_t)(1u << (q >> 5));
                bool ok = !v.empty() && v.size() <= 255 && units == m16 && 5 * v.size() < 32 * (size_t)__builtin_popcount(m16);
                for (int64_t e = 0; ok && e < sel->exc_n[(size_t)k]; e++) { // a listed '=' (BAM code 0) would read as "not listed" on the device
                    const int64_t fr = sel->exc_lo[(size_t)k] + e;
                    if (full->exc_code[fr] == 0 && std::binary_search(v.begin(), v.end(), full->exc_pos[fr])) ok = false;
                }
                if (!ok) continue;
                sel->bl_n[(size_t)k] = (uint8_t)v.size();
                sel->bl_pos.insert(sel->bl_pos.end(), v.begin(), v.end());
                sel->n_bl_units += __builtin_popcount(m16);
                sel->n_seq -= (uint64_t)__builtin_popcount(m16);
            }
            sel->bl_off[(size_t)sel->n_sel] = (int64_t)sel->bl_pos.size();
            for (int64_t k = 0; k < sel->n_sel && !sel->bl_wide; k++) // (two-byte positions as soon as the table holds a read longer than 256 bases: the rule of qlow_pos_wide)
                if (full->l_seq[sel->index[(size_t)k]] > 256) sel->bl_wide = 1;
        }
        if (tuples & 1) { // the small columns as a dictionary: combinations numbered in order of first appearance
            struct KeyHash { size_t operator()(const std::pair<uint64_t, uint32_t> &k) const { return std::hash<uint64_t>()(k.first * 0x9E3779B97F4A7C15ULL + k.second * 0xC2B2AE3D27D4EB4FULL); } };
            std::unordered_map<std::pair<uint64_t, uint32_t>, uint32_t, KeyHash> dict;
            const bool with_um = !sel->umask.empty();
            sel->tup_idx.assign((size_t)sel->n_sel, 0);
            bool ok = true;
            for (int64_t k = 0; k < sel->n_sel && ok; k++) {
                const int64_t i = sel->index[(size_t)k];
                uint32_t aux = sel->bases[(size_t)k] ? full->aux[i] : (full->aux[i] | UZ_AUX_NO_SEQ);
                if (tuples & 2)
                    aux |= uz_cigar_simple_code(full->n_cigar[i], full->n_cigar[i] ? full->cigar[src->coff[(size_t)i]] : 0u, full->l_seq[i]) << UZ_AUX_SIMPLE_SHIFT;
                const uint64_t key = (uint64_t)full->flag[i] | ((uint64_t)full->l_seq[i] << 16) | ((uint64_t)full->n_cigar[i] << 32) |
                                     ((uint64_t)full->mapq[i] << 48) | ((uint64_t)(aux & 0xFFu) << 56);
                const uint8_t low = (tuples & 4) ? sel->n_low[(size_t)k] : (uint8_t)0;
                const uint16_t um16 = with_um ? sel->umask[(size_t)k] : (uint16_t)0;
                const uint8_t nb = sel->bl_n.empty() ? (uint8_t)0 : sel->bl_n[(size_t)k];
                const uint32_t k2 = (uint32_t)low | ((uint32_t)um16 << 8) | ((uint32_t)nb << 24);
                auto it = dict.find({key, k2});
                if (it == dict.end()) {
                    if (dict.size() >= 65536) { ok = false; break; }
                    it = dict.emplace(std::make_pair(key, k2), (uint32_t)dict.size()).first;
                    sel->tup_key.push_back(key); sel->tup_low.push_back(low); sel->tup_um.push_back(um16); sel->tup_nb.push_back(nb);
                }
                sel->tup_idx[(size_t)k] = (uint16_t)it->second;
            }
            if (ok) sel->tuples = tuples;
            else { sel->tup_idx.clear(); sel->tup_key.clear(); sel->tup_low.clear(); sel->tup_um.clear(); sel->tup_nb.clear(); } // more than 65536 combinations: the plain columns
        }
        { // escapes of the 16-bit difference form (counted whether or not the output will use it: cheap)
            const int wk = workers_for(sel->n_sel, threads, 1 << 14);
            std::vector<int64_t> part((size_t)wk + 1, 0);
            std::vector<int64_t> part8((size_t)wk + 1, 0), partn((size_t)wk + 1, 0);
            parallel_slices(sel->n_sel, wk, [&](int64_t a, int64_t b, int slice) {
                int64_t c = 0, c8 = 0, cn = 0;
                int16_t v[4];
                int32_t e[4];
                for (int64_t k = a; k < b; k++) { c += uz_d16_of(sel, k, v, e); c8 += uz_d16_of(sel, k, v, e, true); cn += uz_d16_of(sel, k, v, e, true, true); }
                part[(size_t)slice] = c;
                part8[(size_t)slice] = c8;
                partn[(size_t)slice] = cn;
            });
            for (int k = 0; k < wk; k++) { sel->n_esc16 += part[(size_t)k]; sel->n_esc16_start8 += part8[(size_t)k]; sel->n_esc16_narrow8 += partn[(size_t)k]; }
        }
        { // the pair form: new names by the running maximum of the ids, then every other record's id must be one of them
            const int64_t m = sel->n_sel;
            const int wk = workers_for(m, threads, 1 << 14);
            std::vector<int64_t> smax((size_t)wk + 1, -1), cnt((size_t)wk + 1, 0);
            parallel_slices(m, wk, [&](int64_t a, int64_t b, int slice) {
                int64_t mx = -1;
                for (int64_t k = a; k < b; k++) mx = std::max(mx, (int64_t)full->qname[sel->index[(size_t)k]]);
                smax[(size_t)slice + 1] = mx;
            });
            for (int k = 0; k < wk; k++) smax[(size_t)k + 1] = std::max(smax[(size_t)k + 1], smax[(size_t)k]);
            sel->is_new.assign((size_t)m, 0);
            parallel_slices(m, wk, [&](int64_t a, int64_t b, int slice) {
                int64_t mx = smax[(size_t)slice], c = 0;
                for (int64_t k = a; k < b; k++) {
                    const int64_t q = full->qname[sel->index[(size_t)k]];
                    if (q > mx) { sel->is_new[(size_t)k] = 1; mx = q; c++; }
                }
                cnt[(size_t)slice + 1] = c;
            });
            for (int k = 0; k < wk; k++) cnt[(size_t)k + 1] += cnt[(size_t)k];
            sel->new_ids.assign((size_t)cnt[(size_t)wk], 0);
            parallel_slices(m, wk, [&](int64_t a, int64_t b, int slice) {
                int64_t at = cnt[(size_t)slice];
                for (int64_t k = a; k < b; k++)
                    if (sel->is_new[(size_t)k]) sel->new_ids[(size_t)at++] = full->qname[sel->index[(size_t)k]];
            });
            std::atomic<int> ok{1};
            std::vector<int64_t> part((size_t)wk + 1, 0);
            parallel_slices(m, wk, [&](int64_t a, int64_t b, int) {
                for (int64_t k = a; k < b && ok.load(std::memory_order_relaxed); k++)
                    if (!sel->is_new[(size_t)k] && !std::binary_search(sel->new_ids.begin(), sel->new_ids.end(), full->qname[sel->index[(size_t)k]])) ok.store(0);
            });
            sel->pair8_ok = ok.load();
            if (sel->pair8_ok) {
                parallel_slices(m, wk, [&](int64_t a, int64_t b, int slice) {
                    int64_t c = 0;
                    uint8_t s8, p8;
                    int32_t e[4];
                    bool has[4];
                    for (int64_t k = a; k < b; k++) c += sel_link_of(sel, k, s8, p8, e, has);
                    part[(size_t)slice] = c;
                });
                for (int k = 0; k < wk; k++) sel->n_esc16_pair8 += part[(size_t)k];
            }
        }
        *out = sel;
    });
}

int64_t uz_select_n_records(const uz_select *s) { return s ? s->n_sel : 0; }
int64_t uz_select_n_cigar_total(const uz_select *s) { return s ? (int64_t)s->n_cigar : 0; }
int64_t uz_select_n_row_units(const uz_select *s) { return s ? (int64_t)s->n_units : 0; }
int64_t uz_select_n_seq_units(const uz_select *s) { return s ? (int64_t)s->n_seq : 0; }
int64_t uz_select_n_exc(const uz_select *s) { return s ? s->n_exc : 0; }
int64_t uz_select_n_bl(const uz_select *s) { return (s && !s->bl_n.empty()) ? (int64_t)s->bl_pos.size() : -1; } /* -1: planned without the list form of the bases */
int64_t uz_select_n_bl_units(const uz_select *s) { return s ? s->n_bl_units : 0; }
int uz_select_bl_wide(const uz_select *s) { return s ? s->bl_wide : 0; }
int64_t uz_select_n_qlow_pos(const uz_select *s) { return s ? s->n_qpos : 0; }
int uz_select_end_derivable(const uz_select *s) { return s ? s->end_derivable : 0; }
int64_t uz_select_n_esc16(const uz_select *s) { return s ? s->n_esc16 : 0; }
int64_t uz_select_n_esc16_start8(const uz_select *s) { return s ? s->n_esc16_start8 : 0; }
int64_t uz_select_n_esc16_narrow8(const uz_select *s) { return s ? s->n_esc16_narrow8 : 0; }
int uz_select_pair8_ok(const uz_select *s) { return s ? s->pair8_ok : 0; }
int64_t uz_select_n_esc16_pair8(const uz_select *s) { return (s && s->pair8_ok) ? s->n_esc16_pair8 : -1; }
int64_t uz_select_n_new_names(const uz_select *s) { return s ? (int64_t)s->new_ids.size() : 0; }
int uz_select_qname_map(const uz_select *s, uint32_t *out) {
    if (!s || !out) { last_error = "null argument"; return UZ_IO_E_ARG; }
    if (!s->new_ids.empty()) memcpy(out, s->new_ids.data(), s->new_ids.size() * sizeof(uint32_t));
    return UZ_IO_OK;
}
int64_t uz_select_n_tuples(const uz_select *s) { return (s && s->tuples) ? (int64_t)s->tup_key.size() : -1; }
int64_t uz_select_n_cigar_omitted(const uz_select *s) { return s ? s->n_cigar_simple : 0; }
int uz_select_qlow_pos_wide(const uz_select *s) {
    if (!s) return 0;
    const uz_reads_packed_view *f = &s->src->v;
    if (f->n_low) return f->qlow_pos_wide;
    for (size_t k = 0; k < s->index.size(); k++)
        if (f->l_seq[s->index[k]] > 256) return 1;
    return 0;
}
void uz_select_free(uz_select *s) { delete s; }

int uz_reads_select_fill(const uz_select *s, int threads, uz_reads_packed_view *out, int32_t *orig_index) {
    return guarded([&] {
        if (!s || !out) fail(UZ_IO_E_ARG, "null argument");
        const uz_psrc *src = s->src;
        const uz_reads_packed_view *full = &src->v;
        threads = resolve_threads(threads);
        const int64_t m = s->n_sel;
        out->n_segs = m; out->n_contigs = full->n_contigs; out->min_base_qual = full->min_base_qual; out->n_qnames = full->n_qnames;
        const bool ccompact = out->cigar_compact != 0;
        out->n_cigar_omitted = ccompact ? s->n_cigar_simple : 0;
        out->n_cigar_total = (int64_t)s->n_cigar - out->n_cigar_omitted; out->n_row_units = (int64_t)s->n_units; out->n_seq_units = (int64_t)s->n_seq;
        const bool two_bit = full->seq2 != nullptr;
        if (two_bit && !out->seq2) fail(UZ_IO_E_ARG, "the source table has two-bit base rows: the output view needs seq2 (and the exc_* columns)");
        if (!two_bit && !out->seq4 && s->n_seq) fail(UZ_IO_E_ARG, "the source table has four-bit base rows: the output view needs seq4");
        out->n_exc = two_bit ? s->n_exc : 0;
        if (!out->end && !s->end_derivable) fail(UZ_IO_E_ARG, "the `end` column was left out but a kept record's end is not what its CIGAR gives (uz_select_end_derivable)");
        const bool start8 = out->start_d8 != nullptr;
        const bool narrow8 = out->mate_d8 != nullptr || out->qname_d8 != nullptr;
        const bool pair8 = out->pair_d8 != nullptr;
        const bool d16 = out->start_d != nullptr || start8;
        if (pair8) {
            if (!s->pair8_ok) fail(UZ_IO_E_ARG, "the pair form (pair_d8) is not available for this selection: the name ids of the source do not ascend by first appearance (uz_select_pair8_ok)");
            if (!start8 || narrow8 || out->tlen_s || out->mate_d || out->qname_d) fail(UZ_IO_E_ARG, "pair_d8 comes with start_d8, instead of tlen_s / mate_d* / qname_d*");
            out->n_qnames = (uint32_t)s->new_ids.size(); // the names of the selection, numbered by first appearance (uz_select_qname_map)
        }
        if (start8 && out->start_d) fail(UZ_IO_E_ARG, "start_d and start_d8 are both set");
        if (narrow8 && (!start8 || !out->mate_d8 || !out->qname_d8 || out->mate_d || out->qname_d))
            fail(UZ_IO_E_ARG, "mate_d8 and qname_d8 come together, with start_d8, instead of mate_d / qname_d");
        const int64_t n_esc = pair8 ? s->n_esc16_pair8 : narrow8 ? s->n_esc16_narrow8 : start8 ? s->n_esc16_start8 : s->n_esc16;
        if (d16 && ((!pair8 && (!out->tlen_s || (!narrow8 && (!out->mate_d || !out->qname_d)))) || (n_esc && (!out->esc16_key || !out->esc16_val))))
            fail(UZ_IO_E_ARG, "the 16-bit difference form needs start_d (or start_d8), tlen_s, mate_d, qname_d (or mate_d8, qname_d8) and the esc16_* list");
        out->n_esc16 = d16 ? n_esc : 0;
        std::vector<int64_t> esc_at; // first escape of every slice of the fill loop below
        const int wk_fill = workers_for(m, threads, 4096);
        if (d16) {
            std::vector<int64_t> part((size_t)wk_fill + 1, 0);
            parallel_slices(m, wk_fill, [&](int64_t a, int64_t b, int slice) {
                int64_t c = 0;
                int16_t v[4];
                int32_t e[4];
                uint8_t s8, p8;
                bool has[4];
                for (int64_t k = a; k < b; k++) c += pair8 ? sel_link_of(s, k, s8, p8, e, has) : uz_d16_of(s, k, v, e, start8, narrow8);
                part[(size_t)slice + 1] = c;
            });
            for (int k = 0; k < wk_fill; k++) part[(size_t)k + 1] += part[(size_t)k];
            esc_at = part;
        }
        const bool tup = out->tup != nullptr;
        if (tup) {
            if (!s->tuples) fail(UZ_IO_E_ARG, "the output view asks for the dictionary form (tup) but the selection was planned without it (or met more than 65536 combinations: uz_select_n_tuples)");
            if (((s->tuples & 2) != 0) != (out->cigar_compact != 0) || ((s->tuples & 4) != 0) != (out->tup_n_low != nullptr) || (out->tup_n_low && !out->qlow_pos))
                fail(UZ_IO_E_ARG, "the dictionary of the selection was planned for another output form (cigar_compact / qualities as lists)");
            out->n_tup = (int64_t)s->tup_key.size();
            for (size_t t = 0; t < s->tup_key.size(); t++) {
                const uint64_t key = s->tup_key[t];
                w(out->tup_flag)[t] = (uint16_t)key; w(out->tup_l_seq)[t] = (uint16_t)(key >> 16); w(out->tup_n_cigar)[t] = (uint16_t)(key >> 32);
                w(out->tup_mapq)[t] = (uint8_t)(key >> 48); w(out->tup_aux)[t] = (uint8_t)(key >> 56);
                if (out->tup_n_low) w(out->tup_n_low)[t] = s->tup_low[t];
                if (out->tup_umask) w(out->tup_umask)[t] = s->tup_um[t];
                if (out->tup_n_bl) w(out->tup_n_bl)[t] = s->tup_nb[t];
            }
            if (!s->umask.empty() && !out->tup_umask) fail(UZ_IO_E_ARG, "the selection has unit masks and a dictionary: the output view needs tup_umask");
        } else
            out->n_tup = 0;
        const bool masks = !s->umask.empty();
        if (masks && !out->umask && !(tup && out->tup_umask)) fail(UZ_IO_E_ARG, "the selection was planned with unit masks: the output view needs umask");
        const bool blf = !s->bl_n.empty();
        if (blf != (out->bl_n != nullptr || out->tup_n_bl != nullptr) || (blf && (tup ? !out->tup_n_bl || out->bl_n : !out->bl_n)) ||
            (blf && !s->bl_pos.empty() && (!out->bl_pos || !out->bl_code)) || (blf && s->bl_wide && !out->bl_wide))
            fail(UZ_IO_E_ARG, "the list form of the bases: the output view needs bl_n (or tup_n_bl with the dictionary), bl_pos and bl_code exactly when the selection was planned with it (uz_select_n_bl)");
        out->n_bl = blf ? (int64_t)s->bl_pos.size() : 0; out->n_bl_units = blf ? s->n_bl_units : 0;
        if (blf && out->n_bl) memset(w(out->bl_code), 0, (size_t)(out->n_bl + 3) / 4);
        const bool lists = out->n_low != nullptr || (tup && out->tup_n_low != nullptr);
        if (masks && !lists) fail(UZ_IO_E_ARG, "unit masks need the list form of the qualities in the output (n_low / qlow_pos)");
        if (!lists && !full->qlow) fail(UZ_IO_E_ARG, "the source table has the quality plane as lists: the output view needs n_low / qlow_pos");
        if (lists && !out->qlow_pos_wide && uz_select_qlow_pos_wide(s)) fail(UZ_IO_E_ARG, "reads longer than 256 bases need qlow_pos_wide");
        out->n_qlow_pos = lists ? s->n_qpos : 0;
        std::vector<int64_t> ol;
        if (lists) {
            ol.assign((size_t)m + 1, 0);
            for (int64_t k = 0; k < m; k++)
                ol[(size_t)k + 1] = ol[(size_t)k] + ((s->bases[(size_t)k] && s->n_low[(size_t)k] <= UZ_QLOW_LIST_MAX) ? s->n_low[(size_t)k] : 0);
        }
        std::vector<int64_t> oe;
        if (two_bit) {
            oe.assign((size_t)m + 1, 0);
            for (int64_t k = 0; k < m; k++) oe[(size_t)k + 1] = oe[(size_t)k] + s->exc_n[(size_t)k];
        }
        for (int c = 0; c <= full->n_contigs; c++)
            w(out->contig_off)[c] = std::lower_bound(s->index.begin(), s->index.end(), full->contig_off[c],
                                                     [](int32_t v, int64_t key) { return (int64_t)v < key; }) - s->index.begin();
        for (int c = 0; c < full->n_contigs; c++) {
            int32_t span = 0;
            for (int64_t k = out->contig_off[c]; k < out->contig_off[c + 1]; k++) {
                const int64_t i = s->index[k];
                span = std::max(span, full->end[i] - full->start[i]);
            }
            w(out->max_span)[c] = span;
        }
        // offsets of the kept records in the OUTPUT: prefix sums over the selection
        std::vector<uint64_t> oc((size_t)m + 1, 0), ou((size_t)m + 1, 0), os((size_t)m + 1, 0);
        for (int64_t k = 0; k < m; k++) {
            const int64_t i = s->index[k];
            const bool smp = ccompact && uz_cigar_simple_code(full->n_cigar[i], full->n_cigar[i] ? full->cigar[src->coff[(size_t)i]] : 0u, full->l_seq[i]) != 0;
            oc[k + 1] = oc[k] + (smp ? 0 : full->n_cigar[i]);
            ou[k + 1] = ou[k] + UZ_ROW_UNITS(full->l_seq[i]);
            const uint16_t m16 = masks ? s->umask[(size_t)k] : (uint16_t)UZ_UMASK_ALL;
            os[k + 1] = os[k] + ((s->bases[k] && !(blf && s->bl_n[(size_t)k])) ? (m16 == UZ_UMASK_ALL ? UZ_ROW_UNITS(full->l_seq[i]) : (uint32_t)__builtin_popcount(m16)) : 0);
        }
        int list_mismatch = 0;
        parallel_slices(m, wk_fill, [&](int64_t a, int64_t b, int slice) {
            int64_t esc_next = d16 ? esc_at[(size_t)slice] : 0;
            for (int64_t k = a; k < b; k++) {
                const int64_t i = s->index[k];
                if (out->end) w(out->end)[k] = full->end[i];
                if (pair8) {
                    uint8_t s8, p8;
                    int32_t e[4] = {0, 0, 0, 0};
                    bool has[4];
                    sel_link_of(s, k, s8, p8, e, has);
                    w(out->start_d8)[k] = s8;
                    w(out->pair_d8)[k] = p8;
                    for (int c = 0; c < 4; c++)
                        if (has[c]) {
                            w(out->esc16_key)[esc_next] = ((uint64_t)k << 2) | (uint64_t)c;
                            w(out->esc16_val)[esc_next] = e[c];
                            esc_next++;
                        }
                } else if (d16) {
                    int16_t v[4];
                    int32_t e[4];
                    uz_d16_of(s, k, v, e, start8, narrow8);
                    if (start8) w(out->start_d8)[k] = v[0] == (int16_t)UZ_D16_ESC ? (uint8_t)UZ_D8_ESC : (uint8_t)v[0];
                    else w(out->start_d)[k] = v[0];
                    w(out->tlen_s)[k] = v[1];
                    if (narrow8) { w(out->mate_d8)[k] = (int8_t)v[2]; w(out->qname_d8)[k] = (int8_t)v[3]; }
                    else { w(out->mate_d)[k] = v[2]; w(out->qname_d)[k] = v[3]; }
                    for (int c = 0; c < 4; c++)
                        if (v[c] == (int16_t)((narrow8 && c >= 2) ? UZ_D8S_ESC : UZ_D16_ESC)) {
                            w(out->esc16_key)[esc_next] = ((uint64_t)k << 2) | (uint64_t)c;
                            w(out->esc16_val)[esc_next] = e[c];
                            esc_next++;
                        }
                } else {
                    w(out->start)[k] = full->start[i]; w(out->tlen)[k] = full->tlen[i];
                    const int32_t mt = full->mate[i];
                    int32_t nm = -1;
                    if (mt >= 0 && mt < full->n_segs) { // new index of the mate: its rank in the (sorted) selection
                        auto it = std::lower_bound(s->index.begin(), s->index.end(), mt);
                        if (it != s->index.end() && *it == mt) nm = (int32_t)(it - s->index.begin());
                    }
                    w(out->mate)[k] = nm;
                    w(out->qname)[k] = full->qname[i];
                }
                if (tup) w(out->tup)[k] = s->tup_idx[(size_t)k];
                else {
                    w(out->flag)[k] = full->flag[i]; w(out->l_seq)[k] = full->l_seq[i];
                    w(out->n_cigar)[k] = full->n_cigar[i]; w(out->mapq)[k] = full->mapq[i];
                }
                const uint32_t scode = ccompact ? uz_cigar_simple_code(full->n_cigar[i], full->n_cigar[i] ? full->cigar[src->coff[(size_t)i]] : 0u, full->l_seq[i]) : 0u;
                if (!tup) w(out->aux)[k] = (uint8_t)((s->bases[k] ? full->aux[i] : (full->aux[i] | UZ_AUX_NO_SEQ)) | (scode << UZ_AUX_SIMPLE_SHIFT));
                if (!scode) memcpy(w(out->cigar) + oc[k], full->cigar + src->coff[i], (size_t)full->n_cigar[i] * sizeof(uint32_t));
                const size_t units = UZ_ROW_UNITS(full->l_seq[i]);
                const uint16_t m16 = masks ? s->umask[(size_t)k] : (uint16_t)UZ_UMASK_ALL;
                if (out->umask) w(out->umask)[k] = m16;
                if (!tup && out->bl_n) w(out->bl_n)[k] = blf ? s->bl_n[(size_t)k] : (uint8_t)0;
                if (blf && s->bl_n[(size_t)k]) { // the listed bases: query index + two-bit code (a base that is none of the four: 0, and in the exception list)
                    const uint8_t *row = full->seq2 + (size_t)src->soff[i] * UZ_SEQ2_UNIT_BYTES;
                    for (int64_t e = 0; e < (int64_t)s->bl_n[(size_t)k]; e++) {
                        const int64_t at = s->bl_off[(size_t)k] + e;
                        const uint32_t q = s->bl_pos[(size_t)at];
                        if (out->bl_wide) { w(out->bl_pos)[2 * at] = (uint8_t)(q & 255); w(out->bl_pos)[2 * at + 1] = (uint8_t)(q >> 8); }
                        else w(out->bl_pos)[at] = (uint8_t)q;
                        const uint8_t c2 = (uint8_t)((row[q >> 2] >> (6 - 2 * (q & 3))) & 3u);
                        if (c2) __atomic_fetch_or(w(out->bl_code) + (at >> 2), (uint8_t)(c2 << (2 * (at & 3))), __ATOMIC_RELAXED);
                    }
                    for (int64_t e = 0; e < s->exc_n[(size_t)k]; e++) {
                        const int64_t fr = s->exc_lo[(size_t)k] + e, t2 = oe[(size_t)k] + e;
                        w(out->exc_rec)[t2] = (uint32_t)k; w(out->exc_pos)[t2] = full->exc_pos[fr]; w(out->exc_code)[t2] = full->exc_code[fr];
                    }
                } else if (s->bases[k] && m16 != UZ_UMASK_ALL) { // the staged units only, back to back
                    const size_t ub = two_bit ? UZ_SEQ2_UNIT_BYTES : UZ_SEQ4_UNIT_BYTES;
                    const uint8_t *from = (two_bit ? full->seq2 : full->seq4) + (size_t)src->soff[i] * ub;
                    uint8_t *to = w(two_bit ? out->seq2 : out->seq4) + os[k] * ub;
                    for (size_t u = 0; u < units; u++)
                        if ((m16 >> u) & 1u) { memcpy(to, from + u * ub, ub); to += ub; }
                    if (two_bit)
                        for (int64_t e = 0; e < s->exc_n[(size_t)k]; e++) {
                            const int64_t fr = s->exc_lo[(size_t)k] + e, t2 = oe[(size_t)k] + e;
                            w(out->exc_rec)[t2] = (uint32_t)k; w(out->exc_pos)[t2] = full->exc_pos[fr]; w(out->exc_code)[t2] = full->exc_code[fr];
                        }
                } else if (s->bases[k]) {
                    if (two_bit) {
                        memcpy(w(out->seq2) + os[k] * UZ_SEQ2_UNIT_BYTES, full->seq2 + (size_t)src->soff[i] * UZ_SEQ2_UNIT_BYTES, units * UZ_SEQ2_UNIT_BYTES);
                        for (int64_t e = 0; e < s->exc_n[(size_t)k]; e++) {
                            const int64_t from = s->exc_lo[(size_t)k] + e, to = oe[(size_t)k] + e;
                            w(out->exc_rec)[to] = (uint32_t)k; w(out->exc_pos)[to] = full->exc_pos[from]; w(out->exc_code)[to] = full->exc_code[from];
                        }
                    } else
                        memcpy(w(out->seq4) + os[k] * UZ_SEQ4_UNIT_BYTES, full->seq4 + (size_t)src->soff[i] * UZ_SEQ4_UNIT_BYTES, units * UZ_SEQ4_UNIT_BYTES);
                }
                if (!lists)
                    memcpy(w(out->qlow) + ou[k] * UZ_QLOW_UNIT_BYTES, full->qlow + (size_t)src->uoff[i] * UZ_QLOW_UNIT_BYTES, units * UZ_QLOW_UNIT_BYTES);
                else {
                    if (!tup) w(out->n_low)[k] = s->n_low[(size_t)k];
                    if (s->bases[(size_t)k] && s->n_low[(size_t)k] <= UZ_QLOW_LIST_MAX) {
                        int64_t at = ol[(size_t)k];
                        auto put = [&](int b) {
                            if (out->qlow_pos_wide) { w(out->qlow_pos)[2 * at] = (uint8_t)(b & 255); w(out->qlow_pos)[2 * at + 1] = (uint8_t)(b >> 8); }
                            else w(out->qlow_pos)[at] = (uint8_t)b;
                            at++;
                        };
                        const uint16_t keep16 = masks ? s->umask[(size_t)k] : (uint16_t)UZ_UMASK_ALL;
                        src_low_positions(src, i, [&](int b) { if (unit_kept(keep16, b)) put(b); }); // (the staged units only)
                        if (at != ol[(size_t)k + 1]) __atomic_store_n(&list_mismatch, 1, __ATOMIC_RELAXED);
                    }
                }
                if (orig_index) orig_index[k] = (int32_t)i;
            }
        });
        if (list_mismatch) fail(UZ_IO_E_FORMAT, "listed low-quality positions of the source do not add up to the planned counts");
    });
}

} // extern "C"

// ---- span sums of a packed view (uz_types.h: pk_sums): the host's statement of csrc/k_reads.hip k_off_block_sums + the scan behind it
namespace {
struct SumsView {
    const uz_reads_packed_view *v;
    bool lists;
    int64_t esc_find(int64_t i, int col) const { // value of (record, column) in the escape list, 0 when absent (as the device reads it)
        const uint64_t key = ((uint64_t)i << 2) | (uint64_t)col;
        const uint64_t *k0 = v->esc16_key, *k1 = k0 + v->n_esc16;
        const uint64_t *it = std::lower_bound(k0, k1, key);
        return (it != k1 && *it == key) ? (int64_t)v->esc16_val[it - k0] : 0;
    }
    void vals(int64_t i, uint32_t (&q)[UZ_PK_SUMS]) const {
        uint32_t ls, nc, ax, um, nb;
        int nl;
        if (v->tup) {
            const uint32_t t = v->tup[i];
            ls = v->tup_l_seq[t]; nc = v->tup_n_cigar[t]; ax = v->tup_aux[t];
            nl = lists ? (int)v->tup_n_low[t] : -1;
            um = v->tup_umask ? (uint32_t)v->tup_umask[t] : (v->umask ? (uint32_t)v->umask[i] : UZ_UMASK_ALL);
            nb = v->tup_n_bl ? (uint32_t)v->tup_n_bl[t] : (v->bl_n ? (uint32_t)v->bl_n[i] : 0u);
        } else {
            ls = v->l_seq[i]; nc = v->n_cigar[i]; ax = v->aux[i];
            nl = lists ? (int)v->n_low[i] : -1;
            um = v->umask ? (uint32_t)v->umask[i] : UZ_UMASK_ALL;
            nb = v->bl_n ? (uint32_t)v->bl_n[i] : 0u;
        }
        for (int k = 0; k < UZ_PK_SUMS; k++) q[k] = 0;
        q[4] = (ax & UZ_AUX_SIMPLE_MASK) ? 0u : nc;
        const uint32_t staged = (ax & UZ_AUX_NO_SEQ) ? 0u : (um == UZ_UMASK_ALL ? UZ_ROW_UNITS(ls) : (uint32_t)__builtin_popcount(um));
        q[0] = nc; q[1] = UZ_ROW_UNITS(ls);
        q[2] = nb ? 0u : staged;
        q[7] = nb ? staged : 0u;
        q[8] = nb;
        q[3] = (nl >= 0 && !(ax & UZ_AUX_NO_SEQ) && nl <= UZ_QLOW_LIST_MAX) ? (uint32_t)nl : 0u;
        const bool diff_form = v->tlen_s != nullptr || v->pair_d8 != nullptr;
        if (diff_form) {
            if (v->start_d8) { const uint32_t x = v->start_d8[i]; q[5] = x == UZ_D8_ESC ? (uint32_t)esc_find(i, 0) : x; }
            else { const int x = v->start_d[i]; q[5] = (uint32_t)(x == UZ_D16_ESC ? esc_find(i, 0) : x); }
            if (v->pair_d8) { const uint32_t p = v->pair_d8[i]; q[6] = (p != UZ_P8_SECOND && p != UZ_P8_SECOND_TLEN && p != UZ_P8_OLD) ? 1u : 0u; }
            else if (v->qname_d8) { const int x = v->qname_d8[i]; q[6] = (uint32_t)(x == UZ_D8S_ESC ? esc_find(i, 3) : x); }
            else { const int x = v->qname_d[i]; q[6] = (uint32_t)(x == UZ_D16_ESC ? esc_find(i, 3) : x); }
        }
        if (v->pair_d8) { const uint32_t p = v->pair_d8[i]; q[9] = (p >= 1u && p <= UZ_P8_MAX_DIST) ? 1u : 0u; q[10] = (p == UZ_P8_SECOND || p == UZ_P8_SECOND_TLEN) ? 1u : 0u; }
    }
};
} // namespace

int uz_packed_block_sums(const uz_reads_packed_view *v, int threads, uint64_t *sums, int64_t *n_spans) {
    if (!v || v->n_segs < 0) { last_error = "uz_packed_block_sums: bad view"; return UZ_IO_E_ARG; }
    const int shift = UZ_PK_SHIFT(v->n_segs);
    const int64_t nb = (v->n_segs + ((int64_t)1 << shift) - 1) >> shift;
    if (n_spans) *n_spans = nb;
    if (!sums) return 0;
    return guarded([&] {
        SumsView S{v, v->n_low != nullptr || (v->tup && v->tup_n_low)};
        if (v->n_segs && !(v->tup ? (v->tup_l_seq && v->tup_n_cigar && v->tup_aux) : (v->l_seq && v->n_cigar && v->aux))) fail(UZ_IO_E_ARG, "uz_packed_block_sums: the small columns are missing");
        if ((v->tlen_s || v->pair_d8) && !(v->start_d8 || v->start_d)) fail(UZ_IO_E_ARG, "uz_packed_block_sums: a difference form without its start column");
        parallel_slices(nb, workers_for(nb, resolve_threads(threads), 16), [&](int64_t b0, int64_t b1, int) {
            for (int64_t b = b0; b < b1; b++) {
                uint64_t acc[UZ_PK_SUMS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const int64_t i1 = std::min<int64_t>(v->n_segs, (b + 1) << shift);
                for (int64_t i = b << shift; i < i1; i++) {
                    uint32_t q[UZ_PK_SUMS];
                    S.vals(i, q);
                    for (int k = 0; k < UZ_PK_SUMS; k++) acc[k] += q[k];
                }
                for (int k = 0; k < UZ_PK_SUMS; k++) sums[UZ_PK_SUMS * (b + 1) + k] = acc[k]; // (row b + 1 for now: its own sums; scanned below)
            }
        });
        for (int k = 0; k < UZ_PK_SUMS; k++) sums[k] = 0;
        for (int64_t b = 1; b <= nb; b++)
            for (int k = 0; k < UZ_PK_SUMS; k++) sums[UZ_PK_SUMS * b + k] += sums[UZ_PK_SUMS * (b - 1) + k];
    });
}
