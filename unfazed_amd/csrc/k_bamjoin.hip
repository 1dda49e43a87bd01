// k_bamjoin.hip -- the batch-wide joins of the BAM stage on the device ("BAM record -> SoA observation builder").
//
// What they replace: `bamfile.mate(read)` for every fetched read and every mate of a mate (read_collector.py:400, :185) and the name-keyed tables
// the reference keeps per DNM (read_collector.py:226-234) -- through round 5 the host's plan_finish (csrc/io_stage.cpp) over 64-byte descriptors that
// came DOWN the link, answered by a 32-byte kept list that went back UP.  Here the descriptors never leave HBM: the walk (k_bamwalk.hip) leaves them
// in the batch's slot, and the kept list k_bam_extract unpacks the records from is built where they lie.
//
// The host walks records one after the other, task by task (hash tables in file order, a frontier per generation, a stable sort at the end); a
// device cannot, so the rules are stated ORDER-FREE over all descriptors of the batch (tests/joinmodel.py is this file in numpy; tests/
// test_join_model.py holds it against the one-pass stage on the CPU, tests/test_bamjoin_gpu.py holds these kernels against both):
//
//   join task   of a descriptor: the stage task its walk task is part of (plan column 9) -- or what the host says for a record it walked itself
//               (a task handed back: uz_stage_walk_flagged; a mate looked up through the index: uz_stage_lookup)
//   dropped     the copy a later walk sub-task made of a record the one before met (pos < the stop of the sub-task before); every descriptor of a
//               task the host walked itself
//   mate(x)     among the records of the join task whose reach interval holds x's mate position (covering), or of the look-up task made for x,
//               that carry x's name (two hashes + length) and the other read-of-pair flag and overlap the mate position: the one with the
//               smallest virtual offset.  No record of x's name in that task: the index has to answer (a `need`, handed to the host).
//   members     the fetched records, closed under mate(): level-synchronous rounds over a frontier list; a mate that was not a member joins the
//               next frontier (atomicCAS on its keep word)
//   kept        one record per virtual offset, the copy a fetch returned first: ONE radix sort by (virtual offset, not fetched) -- file order,
//               duplicates of neighbouring tasks folded, records from the index slotted in, all by the same pass
//   name ids    the rank of a name's first kept record among the first kept records (an exclusive scan over "first of my name" flags)
//   offsets     CIGAR words, quality-row units, base-row units, name bytes: the same scan (five sums side by side)
//
// Names lie together through a radix sort of the descriptors by their first name hash: a name's records are a run of two to four entries, and
// every "records of x's name" above is a walk over that run.  Sorts and scans are rocPRIM's (plain library passes); the joins are the kernels here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "uz_bamwalk.h"
#include "uz_ctx.hpp"

namespace {

constexpr uint32_t FPAIRED = 0x1, FMUNMAP = 0x8, FREAD1 = 0x40, FREAD2 = 0x80;
constexpr unsigned long long KEY_NONE = ~0ULL;
enum { C_FRONT0 = 0, C_FRONT1 = 1, C_FRONT2 = 2, C_NEED = 3, C_ERR = 4, C_COUNT = 8 };

struct S5 { // the five running sums of the final pass: first-of-name flags, CIGAR words, row units, base-row units, name bytes
    uint32_t a[5];
};
struct S5Plus {
    __host__ __device__ S5 operator()(const S5 &x, const S5 &y) const {
        S5 r;
#pragma unroll
        for (int k = 0; k < 5; k++) r.a[k] = x.a[k] + y.a[k];
        return r;
    }
};

// a slot of a list for every lane whose `pred` holds: one atomic per wavefront
__device__ __forceinline__ uint32_t wave_slot(int32_t *counter, bool pred) {
    const unsigned long long m = __ballot(pred);
    if (!pred) return 0;
    const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = (uint32_t)atomicAdd(counter, (int32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
}

struct JoinArgs {
    const uz_walk_desc *D;
    int64_t n;
    const int32_t *task;  // the walk plan [UZ_WALK_TASK_COLS n_sub]
    const int32_t *h_flags;
    int32_t *jtask, *keep, *mate, *target;
    unsigned long long *hkey_in;
    uint32_t *hval_in;
    const unsigned long long *hkey;
    const uint32_t *hperm, *inv;
    const int64_t *reach_key;
    const int32_t *reach_a, *reach_host;
    int64_t n_reach;
    int32_t n_ref, n_jt;
    uint32_t *front0, *front1, *need;
    int32_t *cnt;
};

// the tables of the first walk: every reach interval with its reference and stage task (covering), every join task's reference
__global__ void k_join_tables(int n_sub, const int32_t *__restrict__ task, const int32_t *__restrict__ reach, int64_t *reach_key, int32_t *reach_a, int32_t *reach_host,
                              int32_t *jt_tid) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_sub) return;
    const int32_t *tc = task + UZ_WALK_TASK_COLS * (size_t)u;
    jt_tid[tc[9]] = tc[0];
    for (int32_t r = tc[4]; r < tc[5]; r++) {
        reach_key[r] = ((int64_t)tc[0] << 32) + (int64_t)reach[2 * r + 1];
        reach_a[r] = reach[2 * r];
        reach_host[r] = tc[9];
    }
}

// descriptors [i0, n): join task, dropped copies, the state of the closure, the sort key; the fetched records are generation 0
__global__ __launch_bounds__(256) void k_join_init(JoinArgs a, int64_t i0, int first_call) {
    const int64_t i = i0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool direct = false;
    if (i < a.n) {
        const uz_walk_desc &d = a.D[i];
        const uint32_t t = d.task;
        int32_t jt;
        bool drop = false;
        if (t & UZ_WALK_TASK_JOIN) {
            jt = (int32_t)(t & 0x7FFFFFFFu);
            if (jt >= a.n_jt) { a.cnt[C_ERR] = 2; jt = -1; drop = true; }
        } else {
            const int32_t *tc = a.task + UZ_WALK_TASK_COLS * (size_t)t;
            jt = tc[9];
            drop = a.h_flags[jt] != 0 || (t > 0 && tc[9 - UZ_WALK_TASK_COLS] == jt && d.pos < tc[1 - UZ_WALK_TASK_COLS]); // (the sub-task before met it too, and kept it)
        }
        direct = !drop && d.direct != 0;
        a.jtask[i] = drop ? -1 : jt;
        a.keep[i] = direct ? 2 : 0;
        a.mate[i] = -2;
        a.target[i] = -1;
        a.hkey_in[i] = drop ? KEY_NONE : (unsigned long long)d.h1;
        a.hval_in[i] = (uint32_t)i;
    }
    if (first_call) {
        const uint32_t s = wave_slot(a.cnt + C_FRONT0, direct);
        if (direct) a.front0[s] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(256) void k_inverse(int64_t n, const uint32_t *__restrict__ perm, uint32_t *inv) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p < n) inv[perm[p]] = (uint32_t)p;
}

// the stage task whose reach interval holds position pos of reference tid, or -1 (io_stage.cpp: covering)
__device__ __forceinline__ int32_t covering(const JoinArgs &a, int32_t tid, int32_t pos) {
    const int64_t key = ((int64_t)tid << 32) + (int64_t)pos;
    int64_t lo = 0, hi = a.n_reach;
    while (lo < hi) { // the first interval of this reference (or a later one) that ends behind pos
        const int64_t mid = (lo + hi) >> 1;
        if (a.reach_key[mid] <= key) lo = mid + 1; else hi = mid;
    }
    if (lo < a.n_reach && (int32_t)(a.reach_key[lo] >> 32) == tid && a.reach_a[lo] <= pos) return a.reach_host[lo];
    return -1;
}

// One generation of the closure: every member of the frontier looks its mate up; a mate that was no member becomes one and joins the next frontier.
// r: the round (three counters in turn: this round's list, the next one's, the one to clear for the round after)
__global__ __launch_bounds__(256) void k_join_round(JoinArgs a, int r) {
    const int cur = r % 3, nxt = (r + 1) % 3, clr = (r + 2) % 3;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.cnt[clr] = 0;
    const uint32_t *in = (r & 1) ? a.front1 : a.front0;
    uint32_t *out = (r & 1) ? a.front0 : a.front1;
    const int64_t nf = a.cnt[cur];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t f0 = (int64_t)blockIdx.x * 256; f0 < nf; f0 += stride) { // (whole wavefronts stay together: the appends below are ballots)
        const int64_t f = f0 + threadIdx.x;
        bool need = false, joined = false;
        uint32_t i = 0;
        int32_t best = -1;
        if (f < nf) {
            i = in[f];
            if (a.mate[i] == -2) {
                const uz_walk_desc &x = a.D[i];
                const uint32_t fl = x.flag;
                const int32_t mtid = x.mtid, mpos = x.mpos;
                const unsigned long long h = (unsigned long long)x.h1;
                int64_t p = a.inv[i];
                while (p > 0 && a.hkey[p - 1] == h) p--;
                // A copy that will be folded away (the same record met by another task, or found once more through the index, whose other copy is a
                // member already and wins the fold: fetched, or as good and earlier) asks for nothing: only its survivor's mate is ever read.  Without
                // this rule two mates outside every reach interval look each other up for ever -- every answer of the index is a NEW copy of the
                // partner, which asks for a new copy of the first (the host's loop runs into its cap of 64 generations there).
                bool folds = false;
                {
                    const unsigned long long v = x.voff;
                    const int32_t ki = a.keep[i];
                    for (int64_t q = p; q < a.n && a.hkey[q] == h; q++) {
                        const uint32_t e = a.hperm[q];
                        if (e == i || a.D[e].voff != v) continue;
                        const int32_t ke = a.keep[e];
                        if (ke > ki || (ke == ki && ke != 0 && e < i)) folds = true;
                    }
                }
                if (folds || !((fl & FPAIRED) && !(fl & FMUNMAP) && mtid >= 0 && mtid < a.n_ref)) a.mate[i] = -1; // (io_stage.cpp: wants_mate)
                else {
                    const int32_t tgt = a.target[i];
                    const int32_t tc = tgt >= 0 ? tgt : covering(a, mtid, mpos);
                    if (tc < 0) need = true;
                    else {
                        const uint32_t h2 = x.h2, want = (fl ^ (FREAD1 | FREAD2)) & (FREAD1 | FREAD2);
                        const uint8_t ln = x.l_name;
                        bool seen = false;
                        unsigned long long best_v = KEY_NONE;
                        for (; p < a.n && a.hkey[p] == h; p++) {
                            const uint32_t e = a.hperm[p];
                            if (a.jtask[e] != tc) continue;
                            const uz_walk_desc &y = a.D[e];
                            if (y.h2 != h2 || y.l_name != ln) continue;
                            seen = true;
                            if ((int64_t)y.pos < (int64_t)mpos + 1 && (int64_t)y.end > (int64_t)mpos && ((uint32_t)y.flag & want) != 0u && y.voff < best_v) { // (is_mate_of)
                                best_v = y.voff;
                                best = (int32_t)e;
                            }
                        }
                        if (!seen && tgt < 0) need = true; // the covering task does not hold the name: the index answers
                        else {
                            a.mate[i] = best;
                            if (best >= 0) joined = atomicCAS(&a.keep[best], 0, 1) == 0;
                        }
                    }
                }
            }
        }
        const uint32_t sn = wave_slot(a.cnt + C_NEED, need);
        if (need) a.need[sn] = i;
        const uint32_t sj = wave_slot(a.cnt + nxt, joined);
        if (joined) out[sj] = (uint32_t)best;
    }
}

__global__ __launch_bounds__(256) void k_need_recs(int64_t n, const uint32_t *__restrict__ need, const uz_walk_desc *__restrict__ D, uz_need_rec *out) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = need[k];
    const uz_walk_desc &x = D[i];
    uz_need_rec r;
    r.h1 = x.h1; r.h2 = x.h2; r.l_name = x.l_name; r.mtid = x.mtid; r.mpos = x.mpos; r.who = i; r.pad = 0;
    out[k] = r;
}
// the host's answers: need k is answered by join task ans[k]; the askers are the next frontier
__global__ __launch_bounds__(256) void k_set_targets(int64_t n, const uint32_t *__restrict__ need, const int32_t *__restrict__ ans, int32_t n_jt, int32_t *target, uint32_t *front,
                                                     int32_t *cnt) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int32_t t = ans[k];
    if (t < 0 || t >= n_jt) { cnt[C_ERR] = 3; return; }
    target[need[k]] = t;
    front[k] = need[k];
}

// ---- the kept records
struct FinalArgs {
    const uz_walk_desc *D;
    int64_t n;
    const int32_t *jtask, *keep, *mate, *jt_tid;
    const unsigned long long *hkey;
    const uint32_t *hperm, *inv;
    unsigned long long *fkey_in;
    uint32_t *fval_in;
    const unsigned long long *fkey;
    const uint32_t *fidx;
    uint32_t *first, *pos_of_k, *fo, *name_rec;
    const uint32_t *runid;
    int32_t *gidx;
    S5 *s5_in;
    const S5 *s5_out;
    uz_kept_rec *kept;
    unsigned long long *ccount;
    int32_t *cspan, *cnt;
    int64_t *totals;
    int32_t all_bases, n_ref;
};

// sort key of the output order: the virtual offset, a fetched copy in front of a copy that is only somebody's mate; records that are no members last
__global__ __launch_bounds__(256) void k_final_keys(FinalArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int32_t k = a.keep[i];
    unsigned long long key = KEY_NONE;
    if (k != 0) {
        const unsigned long long v = a.D[i].voff;
        if (v >> 63) a.cnt[C_ERR] = 4; // (a file beyond 2^47 bytes)
        key = (v << 1) | (k == 2 ? 0ULL : 1ULL);
    }
    a.fkey_in[i] = key;
    a.fval_in[i] = (uint32_t)i;
    a.gidx[i] = -1;
}
__global__ __launch_bounds__(256) void k_final_first(FinalArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n) return;
    const unsigned long long k = a.fkey[p];
    a.first[p] = (k != KEY_NONE && (p == 0 || (a.fkey[p - 1] >> 1) != (k >> 1))) ? 1u : 0u;
}
// every member copy learns the output index of its survivor; every survivor's place in the sorted order is noted
__global__ __launch_bounds__(256) void k_final_gidx(FinalArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n || a.fkey[p] == KEY_NONE) return;
    const uint32_t f = a.first[p], k = a.runid[p] + f - 1u; // (inclusive rank of the run this copy belongs to)
    a.gidx[a.fidx[p]] = (int32_t)k;
    if (f) a.pos_of_k[k] = (uint32_t)p;
}
// a survivor's name: the first kept record that carries it; its sizes
__global__ __launch_bounds__(256) void k_final_names(FinalArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n) return;
    S5 s = {{0u, 0u, 0u, 0u, 0u}};
    if (a.first[p]) {
        const uint32_t i = a.fidx[p], k = a.runid[p];
        const uz_walk_desc &x = a.D[i];
        const unsigned long long h = (unsigned long long)x.h1;
        const uint32_t h2 = x.h2;
        const uint8_t ln = x.l_name;
        uint32_t fo = k;
        int64_t q = a.inv[i];
        while (q > 0 && a.hkey[q - 1] == h) q--;
        for (; q < a.n && a.hkey[q] == h; q++) {
            const uint32_t e = a.hperm[q];
            const int32_t g = a.gidx[e];
            if (g < 0 || (uint32_t)g >= fo) continue;
            const uz_walk_desc &y = a.D[e];
            if (y.h2 == h2 && y.l_name == ln) fo = (uint32_t)g;
        }
        a.fo[p] = fo;
        const uint32_t units = UZ_ROW_UNITS(x.l_seq);
        const bool bases = a.all_bases || ((a.fkey[p] & 1ULL) == 0ULL);
        s.a[0] = fo == k ? 1u : 0u; s.a[1] = x.n_cigar; s.a[2] = units; s.a[3] = bases ? units : 0u; s.a[4] = x.l_name;
    }
    a.s5_in[p] = s;
}
// the kept list (uz_kept_rec), the record of every name id, records and longest span per reference
__global__ __launch_bounds__(256) void k_final_out(FinalArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = p < a.n && a.first[p] != 0u;
    int32_t tid = -1, span = 0;
    if (on) {
        const uint32_t i = a.fidx[p], k = a.runid[p];
        const uz_walk_desc &x = a.D[i];
        const S5 at = a.s5_out[p];
        const uint32_t fo = a.fo[p];
        uz_kept_rec o;
        o.src = x.src;
        o.qname = a.s5_out[a.pos_of_k[fo]].a[0];
        const int32_t m = a.mate[i];
        o.mate = m >= 0 ? a.gidx[m] : -1;
        const bool bases = a.all_bases || ((a.fkey[p] & 1ULL) == 0ULL);
        o.cig_off = at.a[1]; o.unit_off = at.a[2]; o.seq_off = bases ? at.a[3] : UZ_KEPT_NO_SEQ; o.name_off = at.a[4];
        a.kept[k] = o;
        if (fo == k) a.name_rec[at.a[0]] = k;
        const int32_t jt = a.jtask[i];
        tid = jt >= 0 ? a.jt_tid[jt] : -1;
        if (tid < 0 || tid >= a.n_ref) { a.cnt[C_ERR] = 5; tid = -1; }
        span = x.end - x.pos;
    }
    // the output is in file order, so a wavefront's records lie on one reference (all but the few that straddle two): one atomic per wavefront
    const unsigned long long m = __ballot(on && tid >= 0);
    if (m) {
        const int leader = __ffsll((long long)m) - 1;
        const int32_t t0 = __shfl(tid, leader, 64);
        const bool same = !(on && tid >= 0) || tid == t0;
        if (__all(same)) {
            int32_t mx = (on && tid >= 0) ? span : 0;
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
            if ((int)(threadIdx.x & 63u) == leader) { atomicAdd(&a.ccount[t0], (unsigned long long)__popcll(m)); atomicMax(&a.cspan[t0], mx); }
        } else if (on && tid >= 0) { atomicAdd(&a.ccount[tid], 1ULL); atomicMax(&a.cspan[tid], span); }
    }
}
__global__ void k_final_totals(FinalArgs a) {
    if (blockIdx.x || threadIdx.x) return;
    const int64_t n = a.n;
    int64_t *t = a.totals;
    for (int k = 0; k < JT_COUNT; k++) t[k] = 0;
    if (n > 0) {
        const S5 s = a.s5_out[n - 1], l = a.s5_in[n - 1];
        t[JT_N] = (int64_t)a.runid[n - 1] + (int64_t)a.first[n - 1];
        t[JT_QNAMES] = (int64_t)s.a[0] + l.a[0]; t[JT_CIGAR] = (int64_t)s.a[1] + l.a[1]; t[JT_UNITS] = (int64_t)s.a[2] + l.a[2];
        t[JT_SEQ_UNITS] = (int64_t)s.a[3] + l.a[3]; t[JT_NAME_BYTES] = (int64_t)s.a[4] + l.a[4];
    }
    t[JT_ERR] = a.cnt[C_ERR];
}
// the sums are 32-bit (the kept list's offsets are): a batch whose CIGAR words, units or name bytes do not fit is refused, not wrapped
__global__ __launch_bounds__(256) void k_final_check(FinalArgs a) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p + 1 >= a.n) return;
    const S5 x = a.s5_out[p], y = a.s5_out[p + 1];
#pragma unroll
    for (int k = 1; k < 5; k++)
        if (y.a[k] < x.a[k]) a.cnt[C_ERR] = 6;
}

// parity / debug: the kept records in output order
__global__ __launch_bounds__(256) void k_join_debug(FinalArgs a, unsigned long long *voff, uint32_t *qname, int32_t *mate, uint8_t *bases) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= a.n || !a.first[p]) return;
    const uint32_t k = a.runid[p];
    voff[k] = a.D[a.fidx[p]].voff;
    qname[k] = a.kept[k].qname;
    mate[k] = a.kept[k].mate;
    bases[k] = a.kept[k].seq_off != UZ_KEPT_NO_SEQ ? 1 : 0;
}

// ---- names of name ids
__global__ __launch_bounds__(256) void k_name_lens(int64_t n_ids, const uint32_t *__restrict__ ids, const uint32_t *__restrict__ name_rec, const uz_kept_rec *__restrict__ kept,
                                                   int64_t n_recs, int64_t names_bytes, uint32_t *len) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_ids) return;
    const uint32_t r = name_rec[ids[k]];
    const int64_t a = kept[r].name_off, b = (int64_t)r + 1 < n_recs ? (int64_t)kept[r + 1].name_off : names_bytes;
    len[k] = (uint32_t)(b - a);
}
__global__ __launch_bounds__(256) void k_name_gather(int64_t n_ids, const uint32_t *__restrict__ ids, const uint32_t *__restrict__ name_rec, const uz_kept_rec *__restrict__ kept,
                                                     const uint8_t *__restrict__ names, const uint32_t *__restrict__ len, const uint32_t *__restrict__ off, uint8_t *out) {
    const int64_t k = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; // eight lanes per name
    const int l = threadIdx.x & 7;
    if (k >= n_ids) return;
    const uint8_t *s = names + kept[name_rec[ids[k]]].name_off;
    for (uint32_t b = (uint32_t)l; b < len[k]; b += 8) out[(size_t)off[k] + b] = s[b];
}

inline unsigned grid_of(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 255) / 256); }

template <typename T>
void grow(uz_ctx *c, DevBuf<T> &b, size_t n, int kind, size_t used = 0, hipStream_t st = nullptr) { uz_walk_grow(c, b, n, kind, used, st); }

void sort_pairs(uz_ctx *c, uz_ctx::WalkSlot &w, hipStream_t st, const unsigned long long *kin, unsigned long long *kout, const uint32_t *vin, uint32_t *vout, int64_t n) {
    if (n <= 0) return;
    size_t bytes = 0;
    UZ_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, 64u, st));
    grow(c, w.join.tmp, bytes + 256, 40);
    UZ_HIP(rocprim::radix_sort_pairs((void *)w.join.tmp.p, bytes, kin, kout, vin, vout, (size_t)n, 0u, 64u, st));
}

JoinArgs join_args(uz_ctx::WalkSlot &w) {
    auto &J = w.join;
    JoinArgs a;
    a.D = w.desc_kept.p; a.n = J.n_all; a.task = w.task.p; a.h_flags = J.h_flags.p;
    a.jtask = J.jtask.p; a.keep = J.keep.p; a.mate = J.mate.p; a.target = J.target.p;
    a.hkey_in = J.hkey_in.p; a.hval_in = J.hval_in.p; a.hkey = J.hkey.p; a.hperm = J.hperm.p; a.inv = J.inv.p;
    a.reach_key = J.reach_key.p; a.reach_a = J.reach_a.p; a.reach_host = J.reach_host.p; a.n_reach = 0;
    a.n_ref = J.n_ref; a.n_jt = J.n_host + J.n_look;
    a.front0 = J.front0.p; a.front1 = J.front1.p; a.need = J.need.p; a.cnt = J.cnt.p;
    return a;
}

} // namespace

void uz_join_run(uz_ctx *c, uz_ctx::WalkSlot &w, const JoinPlanHost &P, const uz_walk_desc *x, int64_t n_x, const uint8_t *xaux, int64_t xaux_bytes,
                 const int32_t *look_tid, int64_t n_look, const int32_t *need_jtask) {
    auto &J = w.join;
    hipStream_t st = w.s0;
    const bool first_call = !J.started;
    UZ_REQUIRE(!J.done, UZ_E_STATE, "uz_bam_join: the joins of this batch are finished");
    UZ_REQUIRE(n_x >= 0 && xaux_bytes >= 0 && n_look >= J.n_look && (n_x == 0 || x) && (xaux_bytes == 0 || xaux) && (n_look == 0 || look_tid), UZ_E_ARG, "uz_bam_join: bad arguments");
    int64_t n_reach = 0;
    if (first_call) {
        UZ_REQUIRE(P.n_host >= 0 && P.n_ref >= 0 && (P.n_host == 0 || P.h_flags), UZ_E_ARG, "uz_bam_join: bad plan");
        J.n_host = P.n_host; J.n_ref = P.n_ref; J.all_bases = P.all_bases; J.n_look = 0; J.round = 0; J.n_need = 0; J.aux_bytes = 0;
        J.n_dev = w.n_desc; J.n_all = w.n_desc;
    } else
        UZ_REQUIRE((J.n_need > 0) == (need_jtask != nullptr), UZ_E_ARG, "uz_bam_join: the answers to the last call's needs (uz_stage_lookup), and only then");
    const int64_t n_old = J.n_all, n_new = n_old + n_x;
    UZ_REQUIRE(n_new < (int64_t)0x7FFFFFF0, UZ_E_RANGE, "uz_bam_join: more than 2^31 descriptors");
    // ---- room
    const size_t N = (size_t)n_new + 64;
    const size_t carry = first_call ? 0 : (size_t)n_old; // (what a grown buffer takes along: nothing is in them before the first call)
    grow(c, w.desc_kept, N, 41, carry, st);
    grow(c, J.jtask, N, 42, carry, st); grow(c, J.keep, N, 43, carry, st); grow(c, J.mate, N, 44, carry, st); grow(c, J.target, N, 45, carry, st);
    grow(c, J.hkey_in, N, 46, carry, st); grow(c, J.hval_in, N, 47, carry, st);
    grow(c, J.hkey, N, 48); grow(c, J.hperm, N, 49); grow(c, J.inv, N, 50);
    grow(c, J.front0, N, 51, carry, st); grow(c, J.front1, N, 52, carry, st); grow(c, J.need, N, 53, carry, st);
    grow(c, J.cnt, C_COUNT + 8, 54, first_call ? 0 : (size_t)C_COUNT, st);
    grow(c, J.aux, (size_t)(J.aux_bytes + xaux_bytes) + 1024, 55, (size_t)J.aux_bytes, st);
    grow(c, J.jt_tid, (size_t)(J.n_host + n_look) + 64, 56, first_call ? 0 : (size_t)(J.n_host + J.n_look), st);
    if (first_call) {
        int64_t nr = 0; // reach intervals of the plan: the last walk task's column 5
        // (the plan's arrays were checked by uz_bam_walk; their sizes are the slot's)
        nr = (int64_t)w.n_reach;
        n_reach = nr;
        grow(c, J.reach_key, (size_t)nr + 64, 57); grow(c, J.reach_a, (size_t)nr + 64, 58); grow(c, J.reach_host, (size_t)nr + 64, 59);
        grow(c, J.h_flags, (size_t)J.n_host + 64, 60);
        UZ_HIP(hipMemsetAsync(J.cnt.p, 0, (C_COUNT + 8) * sizeof(int32_t), st));
        if (J.n_host) UZ_HIP(hipMemcpyAsync(J.h_flags.p, P.h_flags, (size_t)J.n_host * 4, hipMemcpyHostToDevice, st));
        // the filtered descriptors (uz_bam_walk counted them): filled here, in walk-task order
        if (w.n_tasks) uz_launch_desc_filter(c, st, true, w.n_tasks, w.desc.p, w.first.p, w.count.p, w.task.p, w.tab_first.p, w.tab.p, w.kcount.p, w.kfirst.p, w.desc_kept.p);
        if (w.n_tasks) hipLaunchKernelGGL(k_join_tables, dim3((unsigned)((w.n_tasks + 127) / 128)), dim3(128), 0, st, (int)w.n_tasks, (const int32_t *)w.task.p, (const int32_t *)w.reach.p,
                                          J.reach_key.p, J.reach_a.p, J.reach_host.p, J.jt_tid.p);
        J.started = true;
    } else
        n_reach = (int64_t)w.n_reach;
    // ---- what the host walked since the last call
    if (n_x) UZ_HIP(hipMemcpyAsync(w.desc_kept.p + n_old, x, (size_t)n_x * sizeof(uz_walk_desc), hipMemcpyHostToDevice, st));
    if (xaux_bytes) UZ_HIP(hipMemcpyAsync(J.aux.p + J.aux_bytes, xaux, (size_t)xaux_bytes, hipMemcpyHostToDevice, st));
    if (n_look > J.n_look) UZ_HIP(hipMemcpyAsync(J.jt_tid.p + J.n_host + J.n_look, look_tid + J.n_look, (size_t)(n_look - J.n_look) * 4, hipMemcpyHostToDevice, st));
    J.n_all = n_new; J.aux_bytes += xaux_bytes; J.n_look = (int32_t)n_look;
    JoinArgs a = join_args(w);
    a.n_reach = n_reach;
    const int64_t i0 = first_call ? 0 : n_old;
    if (n_new > i0) hipLaunchKernelGGL(k_join_init, dim3(grid_of(n_new - i0)), dim3(256), 0, st, a, i0, first_call ? 1 : 0);
    if (first_call || n_x) { // names together: the descriptors by their first name hash (stable: a name's records stay in task / file order)
        sort_pairs(c, w, st, J.hkey_in.p, J.hkey.p, J.hval_in.p, J.hperm.p, n_new);
        if (n_new) hipLaunchKernelGGL(k_inverse, dim3(grid_of(n_new)), dim3(256), 0, st, n_new, (const uint32_t *)J.hperm.p, J.inv.p);
    }
    if (!first_call && J.n_need) { // the answers: the askers are this round's frontier
        DevBuf<int32_t> &ans = J.look_tid; // (scratch of the right type)
        grow(c, ans, (size_t)J.n_need + 64, 61);
        UZ_HIP(hipMemcpyAsync(ans.p, need_jtask, (size_t)J.n_need * 4, hipMemcpyHostToDevice, st));
        const int cur = J.round % 3;
        hipLaunchKernelGGL(k_set_targets, dim3(grid_of(J.n_need)), dim3(256), 0, st, J.n_need, (const uint32_t *)J.need.p, (const int32_t *)ans.p, a.n_jt, J.target.p,
                           (J.round & 1) ? J.front1.p : J.front0.p, J.cnt.p);
        int32_t set[2] = {(int32_t)J.n_need, 0};
        UZ_HIP(hipMemcpyAsync(J.cnt.p + cur, &set[0], 4, hipMemcpyHostToDevice, st));
        UZ_HIP(hipMemcpyAsync(J.cnt.p + C_NEED, &set[1], 4, hipMemcpyHostToDevice, st));
        UZ_HIP(hipStreamSynchronize(st)); // (`set` and the caller's arrays are pageable)
    }
    // ---- the closure: generations until the frontier is empty (four per look at the counters: fetched records, their mates, an empty round)
    const unsigned gr = (unsigned)std::min<int64_t>(std::max<int64_t>(1, (n_new + 255) / 256), 2048);
    int32_t cnt[C_COUNT];
    for (int guard = 0; guard < 64; guard++) {
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_join_round, dim3(gr), dim3(256), 0, st, a, J.round + k);
        J.round += 4;
        UZ_HIP(hipMemcpyAsync(cnt, J.cnt.p, sizeof(cnt), hipMemcpyDeviceToHost, st));
        UZ_HIP(hipStreamSynchronize(st));
        UZ_REQUIRE(cnt[C_ERR] == 0, UZ_E_RANGE, cnt[C_ERR] == 2 ? "uz_bam_join: a descriptor of the host names a join task that does not exist"
                                                                  : "uz_bam_join: an answer names a join task that does not exist");
        if (cnt[J.round % 3] == 0) break;
    }
    J.n_need = cnt[C_NEED];
    if (J.n_need) { // the host's turn (uz_join_needs -> uz_stage_lookup)
        grow(c, J.need_rec, (size_t)J.n_need + 64, 62);
        hipLaunchKernelGGL(k_need_recs, dim3(grid_of(J.n_need)), dim3(256), 0, st, J.n_need, (const uint32_t *)J.need.p, (const uz_walk_desc *)w.desc_kept.p, J.need_rec.p);
        return;
    }
    // ---- the kept records, their order, ids and offsets
    const int64_t n = n_new;
    grow(c, J.fkey_in, N, 63); grow(c, J.fkey, N, 64); grow(c, J.fval_in, N, 65); grow(c, J.fidx, N, 66);
    grow(c, J.first, N, 67); grow(c, J.runid, N, 68); grow(c, J.pos_of_k, N, 69); grow(c, J.fo, N, 70); grow(c, J.gidx, N, 71);
    grow(c, J.s5_in, N * sizeof(S5), 72); grow(c, J.s5_out, N * sizeof(S5), 73);
    grow(c, J.kept, N, 74); grow(c, J.name_rec, N, 75);
    grow(c, J.ccount, (size_t)J.n_ref + 64, 76); grow(c, J.cspan, (size_t)J.n_ref + 64, 77); grow(c, J.totals, 64, 78);
    FinalArgs f;
    f.D = w.desc_kept.p; f.n = n; f.jtask = J.jtask.p; f.keep = J.keep.p; f.mate = J.mate.p; f.jt_tid = J.jt_tid.p;
    f.hkey = J.hkey.p; f.hperm = J.hperm.p; f.inv = J.inv.p; f.fkey_in = J.fkey_in.p; f.fval_in = J.fval_in.p; f.fkey = J.fkey.p; f.fidx = J.fidx.p;
    f.first = J.first.p; f.pos_of_k = J.pos_of_k.p; f.fo = J.fo.p; f.name_rec = J.name_rec.p; f.runid = J.runid.p; f.gidx = J.gidx.p;
    f.s5_in = reinterpret_cast<S5 *>(J.s5_in.p); f.s5_out = reinterpret_cast<const S5 *>(J.s5_out.p); f.kept = J.kept.p;
    f.ccount = J.ccount.p; f.cspan = J.cspan.p; f.cnt = J.cnt.p; f.totals = J.totals.p; f.all_bases = J.all_bases ? 1 : 0; f.n_ref = J.n_ref;
    UZ_HIP(hipMemsetAsync(J.ccount.p, 0, ((size_t)J.n_ref + 1) * 8, st));
    UZ_HIP(hipMemsetAsync(J.cspan.p, 0, ((size_t)J.n_ref + 1) * 4, st));
    if (n) {
        hipLaunchKernelGGL(k_final_keys, dim3(grid_of(n)), dim3(256), 0, st, f);
        sort_pairs(c, w, st, J.fkey_in.p, J.fkey.p, J.fval_in.p, J.fidx.p, n);
        hipLaunchKernelGGL(k_final_first, dim3(grid_of(n)), dim3(256), 0, st, f);
        uz_scan_u32(c, st, J.first.p, J.runid.p, n, J.tmp);
        hipLaunchKernelGGL(k_final_gidx, dim3(grid_of(n)), dim3(256), 0, st, f);
        hipLaunchKernelGGL(k_final_names, dim3(grid_of(n)), dim3(256), 0, st, f);
        size_t bytes = 0;
        const S5 zero = {{0u, 0u, 0u, 0u, 0u}};
        UZ_HIP(rocprim::exclusive_scan(nullptr, bytes, f.s5_in, reinterpret_cast<S5 *>(J.s5_out.p), zero, (size_t)n, S5Plus(), st));
        grow(c, J.tmp, bytes + 256, 40);
        UZ_HIP(rocprim::exclusive_scan((void *)J.tmp.p, bytes, f.s5_in, reinterpret_cast<S5 *>(J.s5_out.p), zero, (size_t)n, S5Plus(), st));
        hipLaunchKernelGGL(k_final_check, dim3(grid_of(n)), dim3(256), 0, st, f);
        hipLaunchKernelGGL(k_final_out, dim3(grid_of(n)), dim3(256), 0, st, f);
    }
    hipLaunchKernelGGL(k_final_totals, dim3(1), dim3(64), 0, st, f);
    UZ_HIP(hipGetLastError());
    std::vector<unsigned long long> cc((size_t)J.n_ref + 1, 0);
    J.max_span_h.assign((size_t)std::max(1, J.n_ref), 0);
    UZ_HIP(hipMemcpyAsync(J.tot_h, J.totals.p, JT_COUNT * 8, hipMemcpyDeviceToHost, st));
    if (J.n_ref) UZ_HIP(hipMemcpyAsync(cc.data(), J.ccount.p, (size_t)J.n_ref * 8, hipMemcpyDeviceToHost, st));
    if (J.n_ref) UZ_HIP(hipMemcpyAsync(J.max_span_h.data(), J.cspan.p, (size_t)J.n_ref * 4, hipMemcpyDeviceToHost, st));
    UZ_HIP(hipStreamSynchronize(st));
    const int64_t e = J.tot_h[JT_ERR];
    UZ_REQUIRE(e == 0, UZ_E_RANGE, e == 4 ? "uz_bam_join: a virtual offset beyond 2^63 (a file of more than 2^47 bytes)"
                                   : e == 5 ? "uz_bam_join: a kept record's join task has no reference"
                                   : e == 6 ? "uz_bam_join: the batch's CIGAR words, row units or name bytes do not fit 32-bit offsets" : "uz_bam_join: inconsistent descriptors");
    J.contig_off_h.assign((size_t)J.n_ref + 1, 0);
    for (int32_t r = 0; r < J.n_ref; r++) J.contig_off_h[(size_t)r + 1] = J.contig_off_h[(size_t)r] + (int64_t)cc[(size_t)r];
    UZ_REQUIRE(J.contig_off_h[(size_t)J.n_ref] == J.tot_h[JT_N], UZ_E_RANGE, "uz_bam_join: the per-reference counts do not add up to the kept records");
    J.done = true;
}

void uz_join_needs(uz_ctx *c, uz_ctx::WalkSlot &w, uz_need_rec *out) {
    auto &J = w.join;
    if (J.n_need == 0) return;
    UZ_HIP(hipMemcpyAsync(out, J.need_rec.p, (size_t)J.n_need * sizeof(uz_need_rec), hipMemcpyDeviceToHost, w.s0));
    UZ_HIP(hipStreamSynchronize(w.s0));
}

void uz_join_fetch(uz_ctx *c, uz_ctx::WalkSlot &w, uint64_t *voff, uint32_t *qname, int32_t *mate, uint8_t *bases, uz_kept_rec *kept) {
    auto &J = w.join;
    UZ_REQUIRE(J.done, UZ_E_STATE, "uz_bam_join_fetch: the joins are not finished");
    const int64_t K = J.tot_h[JT_N], n = J.n_all;
    if (K == 0) return;
    hipStream_t st = w.s0;
    if (kept) UZ_HIP(hipMemcpyAsync(kept, J.kept.p, (size_t)K * sizeof(uz_kept_rec), hipMemcpyDeviceToHost, st));
    if (voff || qname || mate || bases) {
        unsigned long long *dv = nullptr; uint32_t *dq = nullptr; int32_t *dm = nullptr; uint8_t *db = nullptr;
        UZ_HIP(hipMalloc((void **)&dv, (size_t)K * 8)); UZ_HIP(hipMalloc((void **)&dq, (size_t)K * 4)); UZ_HIP(hipMalloc((void **)&dm, (size_t)K * 4)); UZ_HIP(hipMalloc((void **)&db, (size_t)K));
        FinalArgs f;
        memset(&f, 0, sizeof(f));
        f.D = w.desc_kept.p; f.n = n; f.first = J.first.p; f.runid = J.runid.p; f.fidx = J.fidx.p; f.kept = J.kept.p;
        hipLaunchKernelGGL(k_join_debug, dim3(grid_of(n)), dim3(256), 0, st, f, dv, dq, dm, db);
        if (voff) UZ_HIP(hipMemcpyAsync(voff, dv, (size_t)K * 8, hipMemcpyDeviceToHost, st));
        if (qname) UZ_HIP(hipMemcpyAsync(qname, dq, (size_t)K * 4, hipMemcpyDeviceToHost, st));
        if (mate) UZ_HIP(hipMemcpyAsync(mate, dm, (size_t)K * 4, hipMemcpyDeviceToHost, st));
        if (bases) UZ_HIP(hipMemcpyAsync(bases, db, (size_t)K, hipMemcpyDeviceToHost, st));
        UZ_HIP(hipStreamSynchronize(st));
        (void)hipFree(dv); (void)hipFree(dq); (void)hipFree(dm); (void)hipFree(db);
    } else
        UZ_HIP(hipStreamSynchronize(st));
}

void uz_scan_u32(uz_ctx *c, hipStream_t st, const uint32_t *in, uint32_t *out, int64_t n, DevBuf<uint8_t> &tmp) {
    if (n <= 0) return;
    size_t bytes = 0;
    UZ_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
    uz_walk_grow(c, tmp, bytes + 256, 40);
    UZ_HIP(rocprim::exclusive_scan((void *)tmp.p, bytes, in, out, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
}

void uz_launch_name_lens(uz_ctx *c, hipStream_t st, int64_t n_ids, const uint32_t *ids, const uint32_t *name_rec, const uz_kept_rec *kept, int64_t n_recs, int64_t names_bytes,
                         uint32_t *len) {
    if (n_ids <= 0) return;
    hipLaunchKernelGGL(k_name_lens, dim3(grid_of(n_ids)), dim3(256), 0, st, n_ids, ids, name_rec, kept, n_recs, names_bytes, len);
    UZ_HIP(hipGetLastError());
}
void uz_launch_name_gather(uz_ctx *c, hipStream_t st, int64_t n_ids, const uint32_t *ids, const uint32_t *name_rec, const uz_kept_rec *kept, const uint8_t *names,
                           const uint32_t *len, const uint32_t *off, uint8_t *out) {
    if (n_ids <= 0) return;
    hipLaunchKernelGGL(k_name_gather, dim3(grid_of(8 * n_ids)), dim3(256), 0, st, n_ids, ids, name_rec, kept, names, len, off, out);
    UZ_HIP(hipGetLastError());
}
