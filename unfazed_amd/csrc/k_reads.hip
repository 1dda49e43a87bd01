// k_reads.hip -- read stage on gfx950: K3a per-segment QC pre-pass, the sizing pass and the
// per-DNM phase kernel (phase_body.hpp), plus the host-side launch sequence.
//
// Launch shape: a persistent grid of 256-lane workgroups (a few per CU) pulls DNM indices from
// one device counter; each workgroup owns a scratch region in HBM sized from the sizing pass.
// The per-DNM working set (tens of KB) stays L2-resident; scans and small sorts go through LDS.
#include "uz_ctx.hpp"
#include "phase_body.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

// development aid: UZ_TRACE=1 in the environment names every kernel of the read stage before its launch (after a
// stream sync), so a device fault can be attributed
#define UZ_TRACE(name) do { if (getenv("UZ_TRACE")) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "[uz] before %s\n", name); fflush(stderr); } } while (0)

namespace {

// K3a (per-record QC bits) has no kernel of its own any more: with the count of low-quality bases a column of the table, a
// record's QC bits depend on nothing but the record -- the header build (k_pack_rec) writes them once, as a word that
// keeps the mapping quality beside the parameter-independent bits, and the readers apply --min-map-qual (uz_qc_of).

// The sizing pass: one UZ_BW_LANES-lane group per DNM (a DNM has about ten het sites, each costing two lower bounds over its contig's records;
// UZ_BW_SITES items per lane side by side).  What the pass pays for is (a) the length of a DNM's chain of dependent loads and (b) every probe of
// every lane -- about two cycles of its CU each, whether the line is in a cache or not.  So the lanes of a group do TOGETHER what every chain
// would do alike: the first steps of all of a DNM's searches run over the same entries of the index -- its ranges lie within a few thousand
// records of each other -- and take the group one 16-ary search over the coarse index (three steps instead of twelve, 48 probes instead of
// 264) and one over the mid entries under that coarse cell; the UZ_BW_STAGE mid entries from the DNM's lowest bound on go into LDS (coalesced
// loads), where every chain finds its 64-record cell without a probe; what is left per chain is one 32-byte load of the cell's eight mid8
// entries and three steps over the eight record headers under one of them -- ONE line.  A chain beyond the staged entries (a window of more than
// UZ_BW_STAGE x 64 records) takes the levels on its own (uz_lower_bounds_c).
// Round 6, per 100 k DNMs of the bench batch: 0.375 ms (a DNM's window first, its ranges inside it: 43 dependent loads, 120 registers) -> 0.30 (no
// window, mid index) -> 0.22 (32-bit chains: 62 registers, eight waves per SIMD instead of four) -> 0.22 (the group's shared steps: the set-up and
// the shared steps are 0.036 of it -- the rest was the last six steps over the record headers, ~4 lines of HBM traffic per chain) -> 0.150 (mid8).
#ifndef UZ_BW_LANES
#define UZ_BW_LANES 16
#endif
#ifndef UZ_BW_MIN_WAVES
#define UZ_BW_MIN_WAVES 1
#endif
#ifndef UZ_BW_STAGE
#define UZ_BW_STAGE 64
#endif
// first k of [a, b) with ld(k) >= v, else b -- by the GL lanes of a group together (a, b, v uniform over the group; gbase: the group's first lane
// in its wavefront).  Every step: GL probes spread evenly over the bracket, the number of probes below v is the bracket's next GL-th.
template <int GL, typename F>
__device__ __forceinline__ int32_t grp_lower_bound(F ld, int32_t a, int32_t b, int32_t v, int lane, int gbase) {
    constexpr int SH = GL == 64 ? 6 : GL == 32 ? 5 : GL == 16 ? 4 : GL == 8 ? 3 : GL == 4 ? 2 : GL == 2 ? 1 : 0;
    const unsigned long long gmask = GL == 64 ? ~0ull : ((1ull << GL) - 1ull);
    while (a < b) {
        const int32_t chunk = ((b - a) + GL - 1) >> SH;
        const int32_t q = a + lane * chunk;
        const bool below = q < b && ld(q) < v;
        const int cnt = __popcll((__ballot(below) >> gbase) & gmask); // (sorted: the probes below v are the first cnt)
        const int32_t a0 = a;
        if (cnt > 0) a = a0 + (cnt - 1) * chunk + 1;
        if (cnt < GL && a0 + cnt * chunk < b) b = a0 + cnt * chunk;
    }
    return a;
}
__global__ __launch_bounds__(256, UZ_BW_MIN_WAVES) void k_phase_bounds(PhaseArgs a, int32_t *bounds /* [5n] */) {
    constexpr int GL = UZ_BW_LANES, NCH = 2 * UZ_BW_SITES;
    static_assert(GL == 1 || GL == 2 || GL == 4 || GL == 8 || GL == 16 || GL == 32 || GL == 64, "a group is a power of two inside a wavefront");
    __shared__ int32_t s_stage[256 / GL][UZ_BW_STAGE];
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / GL;
    const int lane = threadIdx.x & (GL - 1), gbase = (threadIdx.x & 63) & ~(GL - 1);
    const int d = g < a.n ? (int)g : a.n - 1; // whole groups stay converged for the shuffles below
    long long tp = 0;
    int mh = 0;
    int32_t *b = bounds + 5 * (size_t)d;
    if (g < a.n) {
        const int nc = (int)(a.cand_off[d + 1] - a.cand_off[d]), nh = (int)(a.het_off[d + 1] - a.het_off[d]);
        if (lane == 0) {
            b[0] = b[1] = b[4] = 0;
            b[2] = nh; b[3] = nc;
            for (int k = 0; k < 4; k++) a.pre_win[4 * d + k] = 0;
        }
        const RD &R = a.R;
        const int tid = a.rcontig[d];
        const bool tid_ok = tid >= 0 && tid < R.n_contigs;
        const int32_t clo = tid_ok ? (int32_t)R.contig_off[tid] : 0, chi = tid_ok ? (int32_t)R.contig_off[tid + 1] : 0;
        if (nc > 0 && (!R.mid || !tid_ok || chi - clo <= 128)) { // no index to share: every chain on its own
            uz_phase_bounds_w(a, d, b, lane, GL, tp, mh,
                              [&R, clo, chi](const long long (&v)[NCH], const bool (&on)[NCH], long long (&r)[NCH]) { uz_lower_bounds_c<NCH>(R, clo, chi, v, on, r); });
        } else if (nc > 0) {
            // the DNM's lowest bound: of its own fetch (a point variant's) or of its first het site's (the list is sorted)
            const long long span = R.max_span[tid];
            const bool point = a.vartype[d] == UZ_VT_POINT;
            long long vmin_ = 0x7FFFFFFFLL;
            if (point) {
                const long long position = a.dstart[d];
                vmin_ = ((a.dflags[d] & UZ_DF_FETCH_FALLBACK) ? position : position - 1) - span;
            }
            if (!a.no_extended && nh > 0) {
                const long long v0 = (long long)a.spos[a.het_idx[a.het_off[d]]] - span;
                if (v0 < vmin_) vmin_ = v0;
            }
            const int32_t vmin = uz_clamp_i32(vmin_);
            int32_t lo0 = clo, hi0 = chi;
            if (R.coarse && chi - clo > 8192) {
                const int32_t kl = (clo + 4095) >> 12, kh = chi >> 12;
                const int32_t *cx = R.coarse;
                const int32_t a0 = grp_lower_bound<GL>([cx](int32_t k) { return cx[k]; }, kl, kh > kl ? kh : kl, vmin, lane, gbase);
                if (kh > kl) { lo0 = a0 > kl ? ((a0 - 1) << 12) : clo; hi0 = a0 < kh ? (a0 << 12) : chi; }
            }
            const int32_t *mx = R.mid;
            const int32_t kl_m = (clo + 63) >> 6, kh_m = chi >> 6; // the contig's mid entries
            const int32_t ml = (lo0 + 63) >> 6, mhi = hi0 >> 6;
            int32_t lo1 = lo0; // lower_bound(vmin) is not below lo1
            if (mhi > ml) {
                const int32_t m0 = grp_lower_bound<GL>([mx](int32_t k) { return mx[k]; }, ml, mhi, vmin, lane, gbase);
                if (m0 > ml) lo1 = (m0 - 1) << 6;
            }
            int32_t s0 = (lo1 + 63) >> 6;
            if (s0 < kl_m) s0 = kl_m;
            int32_t s_end = s0 + UZ_BW_STAGE < kh_m ? s0 + UZ_BW_STAGE : kh_m;
            if (s_end < s0) s_end = s0;
            const int32_t cnt = s_end - s0;
            int32_t *sm = s_stage[threadIdx.x / GL];
            for (int i = lane; i < cnt; i += GL) sm[i] = mx[s0 + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the group is inside one wavefront: its LDS accesses are in order; the fences are for the compiler)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool whole = s_end == kh_m; // the staged entries reach the contig's end
            uz_phase_bounds_w(a, d, b, lane, GL, tp, mh, [&R, clo, chi, sm, s0, cnt, lo1, whole](const long long (&v_)[NCH], const bool (&on)[NCH], long long (&r)[NCH]) {
                int32_t x[NCH], y[NCH], v[NCH];
                bool far[NCH], any_far = false;
#pragma unroll
                for (int s = 0; s < NCH; s++) { v[s] = uz_clamp_i32(v_[s]); x[s] = 0; y[s] = on[s] ? cnt : 0; }
                uz_lb_level<NCH>([sm](int32_t k) { return sm[k]; }, x, y, v, 0); // its 64-record cell, in LDS
#pragma unroll
                for (int s = 0; s < NCH; s++) {
                    const int32_t k = x[s];
                    far[s] = on[s] && k == cnt && !whole;
                    any_far |= far[s];
                    x[s] = k > 0 ? ((s0 + k - 1) << 6) : lo1;
                    y[s] = (on[s] && !far[s]) ? (k < cnt ? ((s0 + k) << 6) : chi) : x[s];
                    if (!on[s]) x[s] = y[s] = clo;
                }
                if (R.mid8) uz_mid8_refine<NCH>(R.mid8, x, y, v);
                const RecA *rx = R.ra;
                uz_lb_level<NCH>([rx](int32_t k) { return rx[k].start; }, x, y, v, clo);
#pragma unroll
                for (int s = 0; s < NCH; s++) r[s] = x[s];
                if (any_far) { // beyond the staged entries
                    long long rf[NCH];
                    uz_lower_bounds_c<NCH>(R, clo, chi, v_, far, rf);
#pragma unroll
                    for (int s = 0; s < NCH; s++) if (far[s]) r[s] = rf[s];
                }
            });
        }
    }
#pragma unroll
    for (int o = GL >> 1; o > 0; o >>= 1) {
        tp += __shfl_xor(tp, o, GL);
        const int m2 = __shfl_xor(mh, o, GL);
        mh = m2 > mh ? m2 : mh;
    }
    if (g < a.n && lane == 0) {
        b[1] = (int32_t)(tp > 0x7FFFFFF0LL ? 0x7FFFFFF0LL : tp);
        b[4] = mh;
    }
}

#ifndef UZ_PHASE_PARTS
#define UZ_PHASE_PARTS 8 // XCDs of an MI355X
#endif
#ifndef UZ_PHASE_MIN_WAVES
#define UZ_PHASE_MIN_WAVES 4 // LDS build: a workgroup is ONE wave (wg.hpp), so this is workgroups per SIMD: 4 -> up to 16 DNMs per CU, <= 128 VGPRs (what the LDS arenas leave room for anyway)
#endif
// Two builds of the per-DNM body (phase_body.hpp): k_phase<true> keeps the working arrays of a DNM in its workgroup's LDS
// arena and hands the DNMs that do not fit to k_phase<false>, launched right behind it, which keeps them in HBM scratch.
template <bool LDS>
__global__ __launch_bounds__(WG_NT, LDS ? UZ_PHASE_MIN_WAVES : 5) void k_phase(PhaseArgs a_by_value) {
    __shared__ WgSharedT<LDS ? 1 : WG_SORT_LDS_CAP> sh;
    extern __shared__ __attribute__((aligned(16))) uint8_t uz_lds_arena[];
    // the arguments are read where they lie, phase by phase (phase_body.hpp: uz_args_load); the by-value parameter only gives the kernarg
    // segment its layout (PhaseArgs is the kernel's first and only explicit argument: offset 0)
    (void)a_by_value;
    const PhaseArgsK ap = (PhaseArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    uint8_t *const scr_base = uz_g(ap->scratch) + (size_t)blockIdx.x * ap->scratch_per_wg;
#ifdef UZ_PHASE_TIMING
    if (threadIdx.x < 24) sh.tick[threadIdx.x] = 0;
    struct TickFlush { // on every way out of the kernel
        decltype(sh) *s; unsigned long long *t;
        __device__ ~TickFlush() { __syncthreads(); if (threadIdx.x < 24 && s->tick[threadIdx.x]) atomicAdd(&t[threadIdx.x], s->tick[threadIdx.x]); }
    } tick_flush{&sh, uz_g(ap->timing)};
#endif
    // Work distribution.  A launch over the whole batch: the batch is cut into UZ_PHASE_PARTS contiguous DNM ranges, one cursor each, and a
    // workgroup starts on the range of (blockIdx % PARTS) -- with the usual round-robin placement of workgroups over the 8 XCDs that is
    // "its XCD's range".  DNMs are sorted by position and neighbours share window sites and alignment records, so the lines one of them pulls
    // into the XCD's L2 serve the next (each XCD has its own L2).  A workgroup whose range is exhausted goes on to the other ranges; nothing
    // depends on where a workgroup really runs.  A launch over a list (the DNMs the launch before it gave up): one cursor.
    // (ONE call of the body for both: inlined twice, the kernel was twice the instruction cache's size)
    const bool from_list = ap->from_list != 0;
    int part = from_list ? ap->cursor_slot : (int)(blockIdx.x % UZ_PHASE_PARTS);
    int tried = 0;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int k = atomicAdd(uz_g(ap->work_cursor) + 16 * part, 1); // cursors on separate cache lines
            if (from_list) sh.bcast[0] = k < *uz_g(ap->src_count) ? uz_g(ap->src_list)[k] : -1;
            else {
                const int n = ap->n;
                const int lo = (int)((long long)n * part / UZ_PHASE_PARTS), hi = (int)((long long)n * (part + 1) / UZ_PHASE_PARTS);
                sh.bcast[0] = lo + k < hi ? lo + k : -1;
            }
        }
        __syncthreads();
        const int d = sh.bcast[0];
        if (d < 0) {
            if (from_list || ++tried >= UZ_PHASE_PARTS) break;
            part = (part + 1) % UZ_PHASE_PARTS;
            continue;
        }
        if (uz_phase_dnm<LDS>(ap, scr_base, &sh, LDS ? uz_lds_arena : nullptr, d)) { // (uniform; the HBM build never gives a DNM up)
            if (threadIdx.x == 0) uz_g(ap->retry_list)[atomicAdd(uz_g(ap->retry_count), 1)] = d;
        }
    }
}

struct PhaseState {
    DevBuf<uint8_t> scratch;
    // res: the per-DNM results as they go back to the host in ONE copy -- status [n], counts [4n], origin [n], evidence [n], then (8-byte
    // aligned) the fill level of the list pool; pre_h: first record and length of every het-site fetch range, [2 (n_het + 1)]
    DevBuf<int32_t> bounds, bounds_red, res, cursor, pool, list_len, pre_win, pre_h, retry;
    DevBuf<long long> list_start;
    unsigned long long *pool_cursor = nullptr; // (inside res)
    DevBuf<unsigned int> need_count;
    int32_t *bounds_h = nullptr; // pinned: the copy back must not block the host, the marking kernels follow it
    size_t bounds_h_cap = 0;
    int32_t *res_h = nullptr;    // pinned staging of the per-DNM results
    size_t res_h_cap = 0;
    // the sizes of the batch before this one (with head room): what a batch is first run on (uz_launch_phase)
    bool spec_valid = false;
    Caps spec_caps = {0, 0, 0, 0, 0, 0};
    int spec_arena = 0, spec_arena2 = 0;
    double spec_sumP_per_dnm = 0.0;
    long long spec_runs = 0, spec_misses = 0;
    hipEvent_t bounds_ready = nullptr;
    int n_cus = 0;
    std::vector<long long> list_start_h;
    std::vector<int32_t> list_len_h;
    bool have_lists = false;
    int32_t n = 0;
    // uz_phase_begin left a speculative run in flight: what uz_finish_phase needs to judge it
    bool pending = false;
    Caps pend_caps = {0, 0, 0, 0, 0, 0};
    size_t pend_pool_cap = 0;
    int pend_want_lists = 0;
    bool force_exact = false; // the next run skips the speculative sizing (a batch that outgrew it is run again on its own sizes)
};

__global__ void k_build_coarse(const RecA *ra, int64_t n, int32_t *coarse, int32_t *mid_idx, int32_t *mid8) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((k << 3) < n) {
        const int32_t st = ra[k << 3].start;
        mid8[k] = st;
        if ((k & 7) == 0) mid_idx[k >> 3] = st;
        if ((k & 511) == 0) coarse[k >> 9] = st;
    }
}

// ---- record headers from the staged columns -------------------------------------------------------------
// cigar_off / sq_off are the exclusive prefix sums of n_cigar / UZ_ROW_UNITS(l_seq) over the records: block
// sums, one scan of the block sums, then the pack kernel scans inside its block and writes the headers.
// A span of records is one workgroup's share of the two passes below.  4096 for a table that fills the chip with such spans (a chunk
// of the 100 k-DNM pass: 23 M records, 5.6 k spans; 1024 / 2048 measured slower there: more sums to scan), 1024 for a smaller one --
// the 5 M records of a config-5 chunk are 1.2 k spans of 4096, five 256-lane workgroups per CU, and the pack pass ran at a third of
// its rate per record.
static inline int uz_pk_shift(int64_t n) { return UZ_PK_SHIFT(n); } // (uz_types.h: the packers cut their span sums the same way)
// four running sums per record: CIGAR words, quality-plane units (every record), seq4 units (records with bases), listed
// low-quality positions (list form of the staged plane: records with bases and at most UZ_QLOW_LIST_MAX of them; nl < 0: plane form)
// ... and, fifth, the CIGAR words that travelled (cigar_compact: a record with a simple code owns none); sixth and seventh, the
// differences of start and of the name id to the record before (16-bit difference form: the columns are their running sums,
// modulo 2^32; pair form: the sixth counts the NEW names, a new name's id being the number of new names before it); eighth and
// ninth, the row units and the listed bases of the records whose bases came as a list (bl_*: their units are laid out behind the
// ones that travelled as rows); tenth and eleventh, the FIRST and SECOND records of the pair form (their totals must agree)
#define UZ_PK_SCANNED 9 // the running sums the header build needs per record (the last two are totals only)
// um: which units of the record's rows were staged (UZ_UMASK_ALL: all of them)
__device__ __forceinline__ void pk_vals(uint32_t nc, uint32_t ls, uint32_t aux, int nl, uint32_t um, uint32_t nb, uint32_t (&v)[UZ_PK_SUMS]) {
    v[4] = (aux & UZ_AUX_SIMPLE_MASK) ? 0u : nc;
    v[5] = 0u; v[6] = 0u; v[9] = 0u; v[10] = 0u; // (set by the callers from the difference columns)
    const uint32_t staged = (aux & UZ_AUX_NO_SEQ) ? 0u : (um == UZ_UMASK_ALL ? UZ_ROW_UNITS(ls) : (uint32_t)__popc(um));
    v[0] = nc; v[1] = UZ_ROW_UNITS(ls);
    v[2] = nb ? 0u : staged; // units that travelled as rows
    v[7] = nb ? staged : 0u; // units of a record whose bases came as a list
    v[8] = nb;
    v[3] = (nl >= 0 && !(aux & UZ_AUX_NO_SEQ) && nl <= UZ_QLOW_LIST_MAX) ? (uint32_t)nl : 0u;
}
// the small columns of record i, plain or through the dictionary (uz_reads_packed_view.tup)
struct RecSmall { uint32_t flag, ls, nc, mapq, aux, um; int nl; uint32_t nb; };
// ... of record i whose dictionary index is already at hand
__device__ __forceinline__ RecSmall rec_small_of(const RecColumns &c, int64_t i, uint32_t t) {
    RecSmall r;
    r.flag = c.tup_flag[t]; r.ls = c.tup_l_seq[t]; r.nc = c.tup_n_cigar[t]; r.mapq = c.tup_mapq[t]; r.aux = c.tup_aux[t];
    r.nl = c.lists ? (int)c.tup_n_low[t] : -1;
    r.um = c.tup_umask ? (uint32_t)c.tup_umask[t] : (c.umask ? (uint32_t)c.umask[i] : UZ_UMASK_ALL);
    r.nb = c.tup_n_bl ? (uint32_t)c.tup_n_bl[t] : (c.bl_n ? (uint32_t)c.bl_n[i] : 0u);
    return r;
}
__device__ __forceinline__ RecSmall rec_small(const RecColumns &c, int64_t i) {
    RecSmall r;
    if (c.tup) {
        const uint32_t t = c.tup[i];
        r.flag = c.tup_flag[t]; r.ls = c.tup_l_seq[t]; r.nc = c.tup_n_cigar[t]; r.mapq = c.tup_mapq[t]; r.aux = c.tup_aux[t];
        r.nl = c.lists ? (int)c.tup_n_low[t] : -1;
        r.um = c.tup_umask ? (uint32_t)c.tup_umask[t] : (c.umask ? (uint32_t)c.umask[i] : UZ_UMASK_ALL);
        r.nb = c.tup_n_bl ? (uint32_t)c.tup_n_bl[t] : (c.bl_n ? (uint32_t)c.bl_n[i] : 0u);
    } else {
        r.flag = c.flag[i]; r.ls = c.l_seq[i]; r.nc = c.n_cigar[i]; r.mapq = c.mapq[i]; r.aux = c.aux[i];
        r.nl = c.lists ? (int)c.n_low[i] : -1;
        r.um = c.umask ? (uint32_t)c.umask[i] : UZ_UMASK_ALL;
        r.nb = c.bl_n ? (uint32_t)c.bl_n[i] : 0u;
    }
    return r;
}
// the escape list of the 16-bit difference columns: value of (record, column).  The list is sorted by record, and esc_off (one entry
// per span of records, k_esc_block_off) bounds the search to the handful of entries of the record's own span: a lane that meets
// an escape costs its wave two or three loads instead of a binary search over the whole list
__device__ __forceinline__ int32_t esc16_of(const RecColumns &c, int64_t i, int col) {
    const unsigned long long key = ((unsigned long long)i << 2) | (unsigned long long)col;
    int64_t lo = c.esc_lo, hi = c.esc_hi; // (the entries of the workgroup's own span: set by the kernels below)
    while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (c.esc16_key[mid] < key) lo = mid + 1; else hi = mid; }
    return (lo < c.n_esc16 && c.esc16_key[lo] == key) ? c.esc16_val[lo] : 0; // (a missing entry is caught by the totals / the mate check)
}
// first escape entry at or behind record `rec` (the list is sorted by record): every workgroup of the header build bounds the searches of
// its span with two of these, kept in LDS (a kernel of its own did that for all spans at once: one more small launch per table that had to
// wait for room beside the read stage's persistent workgroups)
__device__ __forceinline__ int64_t esc_lower_bound(const unsigned long long *__restrict__ key, int64_t n_esc, int64_t rec) {
    const unsigned long long want = (unsigned long long)rec << 2;
    int64_t lo = 0, hi = n_esc;
    while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (key[mid] < want) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ uint32_t d16_val(const RecColumns &c, const int16_t *col, int64_t i, int k) {
    const int v = col[i];
    return (uint32_t)(v == UZ_D16_ESC ? esc16_of(c, i, k) : v);
}
// the name-id difference of record i: sixteen bits, or eight (qname_d8); pair form: 1 for a record that brings a new name
__device__ __forceinline__ uint32_t qname_diff(const RecColumns &c, int64_t i) {
    if (c.pair_d8) {
        const uint32_t p = c.pair_d8[i];
        return (p != UZ_P8_SECOND && p != UZ_P8_SECOND_TLEN && p != UZ_P8_OLD) ? 1u : 0u;
    }
    if (c.qname_d8) {
        const int v = c.qname_d8[i];
        return (uint32_t)(v == UZ_D8S_ESC ? esc16_of(c, i, 3) : v);
    }
    return d16_val(c, c.qname_d, i, 3);
}
// the start difference of record i: sixteen bits, or eight (start_d8)
__device__ __forceinline__ uint32_t start_diff(const RecColumns &c, int64_t i) {
    if (c.start_d8) {
        const uint32_t x = c.start_d8[i];
        return x == UZ_D8_ESC ? (uint32_t)esc16_of(c, i, 0) : x;
    }
    return d16_val(c, c.start_d, i, 0);
}
// The form every packer of the product emits (uz_bam_stage_*, uz_reads_select_* with its defaults): the small columns through the dictionary
// with unit masks and counts of low-quality bases, eight-bit start differences, the pair form, compact CIGARs, no `end`.  The header-build
// kernels are compiled once more for exactly that form: the pointers of every other form are known to be null there, their branches fall
// away, and what stays alive across the record loop fits the scalar registers (the general build of k_pack_rec spills 160 of them).
__host__ __device__ inline bool uz_link_form(const RecColumns &c) {
    return c.tup && c.tup_umask && c.tup_n_low && c.lists && c.start_d8 && c.pair_d8 && c.cigar_out && !c.end && !c.plane_in && !c.umask && !c.n_low && !c.bl_n &&
           !c.start && !c.start_d && !c.tlen_s && !c.mate_d8 && !c.qname_d8 && !c.flag;
}
template <bool LINK>
__device__ __forceinline__ RecColumns uz_columns_of(const RecColumns &in) {
    RecColumns c = in;
    if (LINK) {
        c.start = nullptr; c.end = nullptr; c.tlen = nullptr; c.mate = nullptr; c.qname = nullptr;
        c.flag = nullptr; c.l_seq = nullptr; c.n_cigar = nullptr; c.mapq = nullptr; c.aux = nullptr;
        c.umask = nullptr; c.start_d = nullptr; c.tlen_s = nullptr; c.mate_d = nullptr; c.qname_d = nullptr; c.mate_d8 = nullptr; c.qname_d8 = nullptr;
        c.plane_in = nullptr; c.n_low = nullptr; c.bl_n = nullptr; c.lists = 1;
        __builtin_assume(c.tup != nullptr); __builtin_assume(c.tup_umask != nullptr); __builtin_assume(c.tup_n_low != nullptr);
        __builtin_assume(c.start_d8 != nullptr); __builtin_assume(c.pair_d8 != nullptr); __builtin_assume(c.cigar_out != nullptr);
        __builtin_assume(c.cigar_in != nullptr);
    }
    return c;
}
// exclusive scan of the block sums in place by ONE workgroup of 256 lanes; the totals are checked against what the view declared
struct PkWant { unsigned long long cigar, units, seq, qpos, staged /* ~0: not compact */, bl_units, bl; };
__device__ __forceinline__ void scan_block_sums(int64_t nb, unsigned long long *sums, const PkWant &want, int32_t *hflags) {
    __shared__ unsigned long long wpart[UZ_PK_SUMS][4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t chunk = (nb + 255) / 256;
    const int64_t lo = t * chunk < nb ? t * chunk : nb, hi = lo + chunk < nb ? lo + chunk : nb;
    unsigned long long v[UZ_PK_SUMS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, inc[UZ_PK_SUMS];
    for (int64_t i = lo; i < hi; i++)
        for (int k = 0; k < UZ_PK_SUMS; k++) v[k] += sums[UZ_PK_SUMS * i + k];
#pragma unroll
    for (int k = 0; k < UZ_PK_SUMS; k++) { // inclusive scan inside the wave, then across the four waves
        unsigned long long x = v[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = __shfl_up(x, o, 64); if (lane >= o) x += u; }
        inc[k] = x;
        if (lane == 63) wpart[k][wv] = x;
    }
    __syncthreads();
    unsigned long long tot[UZ_PK_SUMS];
#pragma unroll
    for (int k = 0; k < UZ_PK_SUMS; k++) {
        unsigned long long pre = 0, all = 0;
        for (int w = 0; w < 4; w++) { const unsigned long long x = wpart[k][w]; if (w < wv) pre += x; all += x; }
        v[k] = pre + inc[k] - v[k]; // exclusive prefix of this thread's blocks
        tot[k] = all;
    }
    if (t == 0) {
        if (tot[0] != want.cigar || tot[1] != want.units || tot[2] != want.seq || tot[3] != want.qpos || tot[0] > 0xFFFFFFFFULL || tot[1] > 0xFFFFFFFFULL ||
            (want.staged != ~0ULL && tot[4] != want.staged) || tot[7] != want.bl_units || tot[8] != want.bl || tot[1] + tot[7] > 0xFFFFFFFFULL)
            hflags[0] = 1;
        if (tot[9] != tot[10]) hflags[0] = 8; // pair form: as many SECOND records as FIRST ones (k_pair_link checks that they are each other's)
    }
    for (int64_t i = lo; i < hi; i++)
        for (int k = 0; k < UZ_PK_SUMS; k++) {
            const unsigned long long x = sums[UZ_PK_SUMS * i + k];
            sums[UZ_PK_SUMS * i + k] = v[k];
            v[k] += x;
        }
}
template <bool LINK>
__global__ __launch_bounds__(256) void k_off_block_sums(int64_t n, RecColumns c_in, unsigned long long *sums /* [UZ_PK_SUMS nb] */) {
    RecColumns c = uz_columns_of<LINK>(c_in);
    __shared__ unsigned long long part[UZ_PK_SUMS][4];
    __shared__ int64_t esc_span[2];
    const int t = threadIdx.x;
    if (c.n_esc16 > 0 && t < 2) esc_span[t] = esc_lower_bound(c.esc16_key, c.n_esc16, ((int64_t)blockIdx.x + t) << c.pk_shift);
    __syncthreads();
    c.esc_lo = c.n_esc16 > 0 ? esc_span[0] : 0; c.esc_hi = c.n_esc16 > 0 ? esc_span[1] : 0;
    unsigned long long acc[UZ_PK_SUMS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // four records per lane at a time: their column bytes are requested together, then their dictionary entries, then the sums --
    // three round trips to memory for four records instead of three for each
    constexpr int U = 4;
    static_assert(((1 << UZ_PK_SHIFT_SMALL) / 256) % U == 0, "a span is a multiple of 1024 records");
    const int rounds = (1 << c.pk_shift) / 256;
    for (int it = 0; it < rounds; it += U) {
        int64_t idx[U];
        bool in[U];
        RecSmall r[U];
        uint32_t sd[U], pd[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            idx[u] = ((int64_t)blockIdx.x << c.pk_shift) + (it + u) * 256 + t;
            in[u] = idx[u] < n;
            if (!in[u]) idx[u] = n - 1; // (n > 0: the kernel is not launched on an empty table)
        }
        uint32_t tp[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            tp[u] = c.tup ? (uint32_t)c.tup[idx[u]] : 0u;
            sd[u] = c.start_d8 ? (uint32_t)c.start_d8[idx[u]] : 0u;
            pd[u] = c.pair_d8 ? (uint32_t)c.pair_d8[idx[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; u++) r[u] = c.tup ? rec_small_of(c, idx[u], tp[u]) : rec_small(c, idx[u]);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!in[u]) continue;
            const int64_t i = idx[u];
            uint32_t v[UZ_PK_SUMS];
            pk_vals(r[u].nc, r[u].ls, r[u].aux, r[u].nl, r[u].um, r[u].nb, v);
            if (c.diff_form()) {
                v[5] = c.start_d8 ? (sd[u] == UZ_D8_ESC ? (uint32_t)esc16_of(c, i, 0) : sd[u]) : start_diff(c, i);
                v[6] = c.pair_d8 ? ((pd[u] != UZ_P8_SECOND && pd[u] != UZ_P8_SECOND_TLEN && pd[u] != UZ_P8_OLD) ? 1u : 0u) : qname_diff(c, i);
            }
            if (c.pair_d8) { v[9] = (pd[u] >= 1u && pd[u] <= UZ_P8_MAX_DIST) ? 1u : 0u; v[10] = (pd[u] == UZ_P8_SECOND || pd[u] == UZ_P8_SECOND_TLEN) ? 1u : 0u; }
#pragma unroll
            for (int k = 0; k < UZ_PK_SUMS; k++) acc[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < UZ_PK_SUMS; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
        if ((t & 63) == 0) part[k][t >> 6] = acc[k];
    }
    __syncthreads();
    if (t == 0)
        for (int k = 0; k < UZ_PK_SUMS; k++) sums[UZ_PK_SUMS * (size_t)blockIdx.x + k] = part[k][0] + part[k][1] + part[k][2] + part[k][3];
}
// (One workgroup.  Tried in round 4 and dropped: the scan done by whichever workgroup of k_off_block_sums finishes last.  The sums then cross
// between XCDs inside a kernel: with a __threadfence per workgroup every kernel of the step ran 20 - 50 % slower -- two thousand L2
// write-backs per table -- and with agent-scope accesses instead the lone scanning workgroup sat through ~180 uncached round trips.)
__global__ __launch_bounds__(256) void k_off_scan_sums(int64_t nb, unsigned long long *sums, PkWant want, int32_t *hflags) { scan_block_sums(nb, sums, want, hflags); }
// SELF: `sums` holds the raw block sums (no scan kernel ran): every workgroup adds up the sums of the blocks before its own -- a table of a
// few thousand spans costs each workgroup a few microseconds of L2 reads, where the one-workgroup scan kernel between the two passes waited
// ~100 us for room beside the read stage's persistent workgroups; the last workgroup holds the totals against what the view declared.
// pair form: FIRST record i hands its mate j (the SECOND it names) the link back, the name id and the template length
__device__ __forceinline__ void uz_pair_link_one(int64_t i, int64_t j, const uint8_t *__restrict__ pair, const RecA *ra, RecB *rb, int32_t *hflags) {
    if (atomicExch(&rb[j].mate, (int32_t)i) != -2) { hflags[0] = 8; return; } // not a SECOND record, or named twice
    const RecA A = ra[i], M = ra[j];
    if (pair[j] == UZ_P8_SECOND_TLEN) rb[i].tlen = -rb[j].tlen; // (the SECOND brought its own)
    else {
        const int32_t tl = (A.end > M.end ? A.end : M.end) - A.start;
        rb[i].tlen = tl;
        rb[j].tlen = -tl;
    }
    rb[j].qname = rb[i].qname;
}

// diagnostic build only (-DUZ_PACK_TIMING): shader-clock ticks of lane 0 per section of the header build, summed over the workgroups
#ifdef UZ_PACK_TIMING
__device__ unsigned long long uz_pack_ticks[16];
#define PK_TICK(k) do { if (threadIdx.x == 0) { const unsigned long long now__ = __builtin_amdgcn_s_memtime(); ptk__[k] += now__ - ptl__; ptl__ = now__; } } while (0)
#else
#define PK_TICK(k) ((void)0)
#endif
template <bool LINK, bool SELF>
__global__ __launch_bounds__(256) void k_pack_rec(int64_t n, RecColumns c_in, const unsigned long long *__restrict__ sums, PkWant want, RecA *ra, RecB *rb,
                                                  uint32_t *fm, uint32_t *qoff, uint8_t *nlow, uint16_t *umask_out, uint32_t *plane_out,
                                                  uint16_t *qs, int32_t *coarse, int32_t *mid_idx, int32_t *mid8, int32_t *hflags, int host_sums) {
    RecColumns c = uz_columns_of<LINK>(c_in);
    __shared__ uint32_t wsum[UZ_PK_SCANNED][4];
    __shared__ int64_t esc_span[2];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
#ifdef UZ_PACK_TIMING
    unsigned long long ptk__[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptl__ = __builtin_amdgcn_s_memtime();
#endif
    if (c.n_esc16 > 0 && t < 2) esc_span[t] = esc_lower_bound(c.esc16_key, c.n_esc16, ((int64_t)blockIdx.x + t) << c.pk_shift);
    __syncthreads();
    c.esc_lo = c.n_esc16 > 0 ? esc_span[0] : 0; c.esc_hi = c.n_esc16 > 0 ? esc_span[1] : 0;
    unsigned long long run[UZ_PK_SCANNED];
    if constexpr (SELF) {
        __shared__ unsigned long long pre_s[UZ_PK_SUMS][4];
        unsigned long long acc[UZ_PK_SUMS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool last = blockIdx.x + 1 == gridDim.x;
        const int64_t nbk = (int64_t)blockIdx.x + (last ? 1 : 0); // (the last workgroup adds its own block too: the totals)
        for (int64_t b = t; b < nbk; b += 256) {
            unsigned long long row[UZ_PK_SUMS];
#pragma unroll
            for (int k = 0; k < UZ_PK_SUMS; k++) row[k] = sums[UZ_PK_SUMS * (size_t)b + k];
#pragma unroll
            for (int k = 0; k < UZ_PK_SUMS; k++) acc[k] += row[k];
        }
        // (the last workgroup's own block: part of the totals, not of its prefix)
        unsigned long long own[UZ_PK_SUMS];
#pragma unroll
        for (int k = 0; k < UZ_PK_SUMS; k++) own[k] = last ? sums[UZ_PK_SUMS * (size_t)blockIdx.x + k] : 0ULL;
#pragma unroll
        for (int k = 0; k < UZ_PK_SUMS; k++) {
            unsigned long long x = acc[k];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
            if (lane == 0) pre_s[k][wv] = x;
        }
        __syncthreads();
        unsigned long long tot[UZ_PK_SUMS];
#pragma unroll
        for (int k = 0; k < UZ_PK_SUMS; k++) tot[k] = pre_s[k][0] + pre_s[k][1] + pre_s[k][2] + pre_s[k][3];
#pragma unroll
        for (int k = 0; k < UZ_PK_SCANNED; k++) run[k] = tot[k] - own[k];
        if (last && t == 0) {
            if (tot[0] != want.cigar || tot[1] != want.units || tot[2] != want.seq || tot[3] != want.qpos || tot[0] > 0xFFFFFFFFULL || tot[1] > 0xFFFFFFFFULL ||
                (want.staged != ~0ULL && tot[4] != want.staged) || tot[7] != want.bl_units || tot[8] != want.bl || tot[1] + tot[7] > 0xFFFFFFFFULL)
                hflags[0] = 1;
            if (tot[9] != tot[10]) hflags[0] = 8; // pair form: as many SECOND records as FIRST ones (k_pair_link checks that they are each other's)
        }
    } else {
#pragma unroll
        for (int k = 0; k < UZ_PK_SCANNED; k++) run[k] = sums[UZ_PK_SUMS * (size_t)blockIdx.x + k];
    }
    // host_sums: the offsets come from the packer (uz_types.h pk_sums; the host has checked that the rows ascend and end at the declared totals).
    // Every record's variable-length parts must then end in front of the NEXT span's offsets -- nothing is read or written beyond the span's share
    // -- and the span's own sums must add up to exactly that row: a packer that counted wrong is refused, whatever it counted.
    // (the next row sits in LDS: eighteen registers less across the record loop)
    __shared__ unsigned long long nxt[UZ_PK_SCANNED];
    if (t < UZ_PK_SCANNED) nxt[t] = host_sums ? sums[UZ_PK_SUMS * ((size_t)blockIdx.x + 1) + t] : ~0ULL;
    __syncthreads();
    // the dictionary index of the NEXT round's record is requested a round ahead: its table entries can then be fetched as soon as
    // the round begins, instead of after a round trip of their own
    uint32_t tp_next = 0;
    {
        const int64_t i0 = ((int64_t)blockIdx.x << c.pk_shift) + t;
        if (c.tup && i0 < n) tp_next = c.tup[i0];
    }
    const int rounds = (1 << c.pk_shift) / 256;
    PK_TICK(0); // set-up
    for (int it = 0; it < rounds; it++) {
        const int64_t i = ((int64_t)blockIdx.x << c.pk_shift) + it * 256 + t;
        const bool in = i < n;
        const uint32_t tp = tp_next;
        if (c.tup && it + 1 < rounds && i + 256 < n) tp_next = c.tup[i + 256];
        RecSmall rs = {0u, 0u, 0u, 0u, 0u, UZ_UMASK_ALL, c.lists ? 0 : -1, 0u};
        if (in) rs = c.tup ? rec_small_of(c, i, tp) : rec_small(c, i);
        const uint32_t nc = rs.nc, ls = rs.ls, ax = rs.aux;
        const int nl = rs.nl;
        const uint32_t um = rs.um, nb = rs.nb;
        uint32_t v[UZ_PK_SUMS], inc[UZ_PK_SCANNED];
        pk_vals(nc, ls, ax, nl, um, nb, v);
        if (c.diff_form() && in) { v[5] = start_diff(c, i); v[6] = qname_diff(c, i); }
        PK_TICK(1); // small columns + dictionary + differences
#pragma unroll
        for (int k = 0; k < UZ_PK_SCANNED; k++) inc[k] = wv_incl_scan(v[k]); // (the last two sums are totals only; DPP scans: wg.hpp)
        __syncthreads(); // wsum of the previous round has been read
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < UZ_PK_SCANNED; k++) wsum[k][wv] = inc[k];
        }
        __syncthreads();
        uint32_t pre[UZ_PK_SCANNED], tot[UZ_PK_SCANNED];
#pragma unroll
        for (int k = 0; k < UZ_PK_SCANNED; k++) {
            pre[k] = 0; tot[k] = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) { if (w < wv) pre[k] += wsum[k][w]; tot[k] += wsum[k][w]; }
        }
        PK_TICK(2); // scans
        bool fits = true;
        if (host_sums && in) {
#pragma unroll
            for (int k = 0; k < UZ_PK_SCANNED; k++)
                if (k != 5 && k != 6) fits &= run[k] + pre[k] + inc[k] <= nxt[k];
            if (!fits) hflags[0] = 1;
        }
        if (in && fits) {
            RecA A;
            RecB B;
            // base rows: the units that travelled as rows in link order, then the units of the records whose bases came as a list
            const uint32_t sq = (ax & UZ_AUX_NO_SEQ) ? UZ_NO_SEQ_OFF
                                : (nb ? (uint32_t)((unsigned long long)c.n_seq_link + run[7] + pre[7] + inc[7] - v[7]) : (uint32_t)(run[2] + pre[2] + inc[2] - v[2]));
            const uint32_t cg = (uint32_t)(run[0] + pre[0] + inc[0] - v[0]);
            // the four wide columns: plain, or from their 16-bit differences (start and name id: running sums, this record included)
            int32_t st0, tl0, mt0;
            uint32_t qn0;
            if (c.pair_d8) {
                // pair form: a FIRST record knows its mate and its (new) name id, and gets its template length from k_pair_link once
                // every `end` is there; a SECOND record gets all three from the FIRST that names it (mate -2 until then)
                st0 = (int32_t)(uint32_t)(run[5] + pre[5] + inc[5]);
                const uint32_t p = c.pair_d8[i];
                qn0 = (uint32_t)(run[6] + pre[6] + inc[6]) - 1u; // a new name's id = new names before it
                tl0 = 0;
                if (p == UZ_P8_SECOND) { mt0 = -2; qn0 = 0u; }
                else if (p == UZ_P8_SECOND_TLEN) { mt0 = -2; qn0 = 0u; tl0 = esc16_of(c, i, 1); }
                else if (p <= UZ_P8_MAX_DIST) {
                    mt0 = (int32_t)(i + p);
                    if (mt0 >= n) { hflags[0] = 7; mt0 = -1; }
                } else {
                    tl0 = esc16_of(c, i, 1);
                    mt0 = esc16_of(c, i, 2);
                    if (p == UZ_P8_OLD) qn0 = (uint32_t)esc16_of(c, i, 3);
                    if (mt0 < -1 || mt0 >= n) { hflags[0] = 7; mt0 = -1; }
                }
            } else if (c.tlen_s) {
                st0 = (int32_t)(uint32_t)(run[5] + pre[5] + inc[5]);
                qn0 = (uint32_t)(run[6] + pre[6] + inc[6]);
                tl0 = (int32_t)d16_val(c, c.tlen_s, i, 1);
                if (c.mate_d8) {
                    const int md = c.mate_d8[i];
                    mt0 = md == UZ_D8S_NONE ? -1 : (md == UZ_D8S_ESC ? esc16_of(c, i, 2) : (int32_t)(i + md));
                } else {
                    const int md = c.mate_d[i];
                    mt0 = md == UZ_D16_NONE ? -1 : (md == UZ_D16_ESC ? esc16_of(c, i, 2) : (int32_t)(i + md));
                }
                if (mt0 < -1 || mt0 >= n) { hflags[0] = 7; mt0 = -1; }
            } else { st0 = c.start[i]; tl0 = c.tlen[i]; mt0 = c.mate[i]; qn0 = c.qname[i]; }
            PK_TICK(3); // wide columns (pair form / escapes)
            // The record's CIGAR words, fetched ONCE and all together (the first four in one round trip; a short read has one to
            // three): its end and the two counts of the QC word come from registers, and the device's store gets the words -- the
            // travelled ones, or the one a simple record's aux byte names (cigar_compact).
            const uint32_t code = c.cigar_out ? (ax & UZ_AUX_SIMPLE_MASK) >> UZ_AUX_SIMPLE_SHIFT : 0u;
            if (code && nc != 1) hflags[0] = 6;
            const uint32_t *src_w = c.cigar_out ? c.cigar_staged + (run[4] + pre[4] + inc[4] - v[4]) : c.cigar_in + cg;
            const bool have_words = c.cigar_in != nullptr; // (an ASCII upload lays the words out after this kernel: k_pack_ascii sets the two bits)
            uint32_t w4[4] = {0u, 0u, 0u, 0u};
            if (code) w4[0] = uz_cigar_simple_word(code, ls);
            else if (have_words) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) if (k < nc) w4[k] = src_w[k];
            }
            int64_t ref_len = 0;
            int cig_nonmatch = 0, cig_none = 0; // the two CIGAR counts of the QC word
            auto take = [&](uint32_t w) {
                const uint32_t op = w & 15u;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += (int64_t)(w >> 4); // M D N = X
                if (!(ax & UZ_AUX_DECODE_BAD)) uz_cigar_op_counts(w, cig_nonmatch, cig_none);
            };
            if (have_words) {
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) if (k < nc) { take(w4[k]); if (c.cigar_out) c.cigar_out[cg + k] = w4[k]; }
                for (uint32_t k = 4; k < nc; k++) { const uint32_t w = src_w[k]; take(w); if (c.cigar_out) c.cigar_out[cg + k] = w; }
            }
            // (column left out: as htslib's bam_endpos -- uz_bam_endpos in pack.hpp states it for the host)
            const int32_t en0 = c.end ? c.end[i] : (((rs.flag & 4u) || nc == 0 || !have_words) ? st0 + 1 : (int32_t)(st0 + (ref_len > 0 ? ref_len : 1)));
            PK_TICK(4); // CIGAR
            int low_for_qc = 0; // (an ASCII upload has no counts yet: uz_build_qlow sets the bit that depends on them)
            uz_pack_rec(A, B, st0, en0, cg, sq, mt0, qn0, (uint16_t)ls, (uint16_t)nc, tl0);
            if ((i & 7) == 0) { // the search index: start of every 4096th record (coarse), of every 64th (mid) and of every 8th (mid8)
                mid8[i >> 3] = st0;
                if ((i & 63) == 0) {
                    mid_idx[i >> 6] = st0;
                    if ((i & 4095) == 0) coarse[i >> 12] = st0;
                }
            }
            ra[i] = A;
            rb[i] = B;
            fm[i] = uz_pack_fm(rs.flag, rs.mapq, ax);
            const int units = (int)UZ_ROW_UNITS(ls);
            umask_out[i] = (uint16_t)(nb ? (um | UZ_UMASK_LISTED) : um);
            if (um != UZ_UMASK_ALL && (units > 15 || (um >> units) != 0u)) hflags[0] = 5; // a unit beyond the read, or a read too long for a mask
            PK_TICK(5); // header stores
            if (nb) {
                // The listed bases -> the record's rows (one 16-byte row per unit of its mask; bases that were not listed stay code 0, which
                // the readers of a listed record refuse: phase_body.hpp uz_base).  Positions ascend, so the units come in row order.
                if (um == UZ_UMASK_ALL || (ax & UZ_AUX_NO_SEQ)) hflags[0] = 9;
                else {
                    const unsigned long long at = run[8] + pre[8] + inc[8] - v[8];
                    uint32_t w4r[4] = {0u, 0u, 0u, 0u};
                    int cur = -1, last = -1;
                    uint32_t row = sq, seen = 0u;
                    bool bad = false;
                    for (uint32_t e = 0; e < nb; e++) {
                        const unsigned long long g = at + e;
                        const int pos = c.bl_wide ? ((int)c.bl_pos[2 * g] | ((int)c.bl_pos[2 * g + 1] << 8)) : (int)c.bl_pos[g];
                        const uint32_t code = ((uint32_t)c.bl_code[g >> 2] >> (2u * (uint32_t)(g & 3ULL))) & 3u;
                        const int u = pos >> 5;
                        bad |= pos <= last || pos >= (int)ls || u > 14 || !((um >> u) & 1u);
                        last = pos;
                        if (u != cur) {
                            if (cur >= 0) {
#pragma unroll
                                for (int q = 0; q < 4; q++) { c.seq4_out[4 * (size_t)row + q] = w4r[q]; w4r[q] = 0u; }
                                row++;
                            }
                            cur = u;
                            seen |= 1u << (u & 15);
                        }
                        const int k32 = pos & 31; // byte k32 >> 1 of the row, high nibble when k32 is even
                        w4r[k32 >> 3] |= (1u << code) << (8 * ((k32 >> 1) & 3) + ((k32 & 1) ? 0 : 4));
                    }
                    if (cur >= 0) {
#pragma unroll
                        for (int q = 0; q < 4; q++) c.seq4_out[4 * (size_t)row + q] = w4r[q];
                    }
                    if (bad || seen != um) hflags[0] = 9; // a position outside the mask / not ascending, or a unit of the mask without a listed base
                }
            }
            PK_TICK(6); // listed bases
            if (nl >= 0) {
                // list form of the staged plane: the count as it is; a quality row (at the record's base-row position) only for a
                // record whose bits can be asked for, written here from its listed positions
                nlow[i] = (uint8_t)nl;
                low_for_qc = nl;
                const bool listed = v[3] == (uint32_t)nl && !(ax & UZ_AUX_NO_SEQ) && nl <= UZ_QLOW_LIST_MAX;
                qoff[i] = listed ? sq : UZ_NO_QLOW_OFF;
                if (listed) {
                    const unsigned long long at = run[3] + pre[3] + inc[3] - v[3];
                    int pos[UZ_QLOW_LIST_MAX];
                    bool bad = false;
#pragma unroll
                    for (int e = 0; e < UZ_QLOW_LIST_MAX; e++) {
                        pos[e] = -1;
                        if (e < nl) {
                            pos[e] = c.qpos_wide ? ((int)c.qlow_pos[2 * (at + e)] | ((int)c.qlow_pos[2 * (at + e) + 1] << 8)) : (int)c.qlow_pos[at + e];
                            bad |= pos[e] >= (int)ls || (e > 0 && pos[e] <= pos[e - 1]);
                        }
                    }
                    if (bad) hflags[0] = 4;
                    uint32_t at_row = sq; // the quality row holds the same units as the base row: the staged ones
                    if (um != UZ_UMASK_ALL) { // (usually one or two of a read's five: one round per staged unit)
                        uint32_t m = um;
                        while (m) {
                            const int u = __ffs((int)m) - 1;
                            m &= m - 1u;
                            uint32_t w = 0;
#pragma unroll
                            for (int e = 0; e < UZ_QLOW_LIST_MAX; e++)
                                if (pos[e] >= 0 && (pos[e] >> 5) == u) w |= 1u << (pos[e] & 31);
                            plane_out[(size_t)at_row++] = w;
                        }
                    } else
                        for (int u = 0; u < units; u++) {
                            uint32_t w = 0;
#pragma unroll
                            for (int e = 0; e < UZ_QLOW_LIST_MAX; e++)
                                if (pos[e] >= 0 && (pos[e] >> 5) == u) w |= 1u << (pos[e] & 31);
                            plane_out[(size_t)at_row++] = w;
                        }
                }
            } else {
                const uint32_t qo = (uint32_t)(run[1] + pre[1] + inc[1] - v[1]);
                qoff[i] = qo;
                if (c.plane_in) { // the plane itself was staged: count its bits once
                    int low = 0;
                    for (int u = 0; u < units; u++) {
                        uint32_t w = c.plane_in[(size_t)qo + u];
                        const int valid = (int)ls - 32 * u;
                        if (valid < 32) w &= valid > 0 ? ((1u << valid) - 1u) : 0u;
                        low += __popc(w);
                    }
                    nlow[i] = (uint8_t)(low > 255 ? 255 : low);
                    low_for_qc = low;
                }
            }
            qs[i] = uz_qs_word(rs.flag, ax, rs.mapq, low_for_qc, (int)nc, cig_nonmatch, cig_none);
            PK_TICK(7); // quality list + QC word
        }
#pragma unroll
        for (int k = 0; k < UZ_PK_SCANNED; k++) run[k] += tot[k];
    }
    PK_TICK(8);
    if (c.pair_d8) {
        // the pairs that lie inside this span are joined here, by the workgroup that has just written both headers (they sit in its L2 lines); a FIRST
        // whose SECOND belongs to the next span is left to k_pair_link, which then looks at the last 256 records of every span only
        __syncthreads();
        const int64_t s0 = (int64_t)blockIdx.x << c.pk_shift, s1 = s0 + ((int64_t)1 << c.pk_shift);
        for (int it = 0; it < rounds; it++) {
            const int64_t i = s0 + it * 256 + t;
            if (i >= n) break;
            const uint32_t p = c.pair_d8[i];
            if (p < 1u || p > UZ_P8_MAX_DIST) continue;
            const int64_t j = i + p;
            if (j >= n || j >= s1) continue; // (beyond the table: flagged above; in the next span: k_pair_link)
            uz_pair_link_one(i, j, c.pair_d8, ra, rb, hflags);
        }
    }
    PK_TICK(9); // pairs inside the span
#ifdef UZ_PACK_TIMING
    if (threadIdx.x == 0) for (int k = 0; k < 10; k++) atomicAdd(&uz_pack_ticks[k], ptk__[k]);
#endif
    if (host_sums && t == 0) {
        bool same = true;
#pragma unroll
        for (int k = 0; k < UZ_PK_SCANNED; k++) same &= (k == 5 || k == 6) ? (uint32_t)run[k] == (uint32_t)nxt[k] : run[k] == nxt[k]; // (the two difference columns count modulo 2^32)
        if (!same) hflags[0] = 1;
    }
}

// ---- the header build of the link form, from LDS -----------------------------------------------------------------------------------------
// k_pack_rec reads a record's parts where they lie: per round of 256 records a dozen dependent trips to memory (dictionary entries, escape
// searches, CIGAR words, listed positions one at a time, quality lists, and a second pass over the span for its pairs) -- 30 k cycles per
// round, 1.0 ms per 9.9 M-record chunk of the staged pass against a streaming bound of 0.12 ms (profiles/r05: section timing).  The form
// every packer of the product emits comes with the span sums of the packer (pk_sums): a workgroup knows before it starts where every
// variable-length list of its span begins and ends.  So it stages ALL of its span's input -- the three byte columns, the dictionary, its
// share of the quality lists, listed bases, travelled CIGAR words and escapes, ~6 KB -- into LDS with one round of coalesced loads, and
// decodes from there; the pairs of a span are joined through LDS too (a FIRST record tells its SECOND how far back it stands), so a header is
// stored once, finished.  Only a FIRST whose SECOND lies in the next span is left to k_pair_link.
// Taken for: the link form (uz_link_form) with host sums, spans of 1024 records, byte positions.  The dictionary of the small columns (a few
// thousand combinations of eight values) is packed into one 16-byte entry per combination first (k_pack_dict): a record's small columns are one load.
// A list of a span that outgrows its LDS room is read from memory by that workgroup (slower, same result).
#define UZ_PL_SPAN 1024
#define UZ_PL_QP 4096   // staged quality-list bytes per span
#define UZ_PL_BL 4096   // staged listed-base positions per span
#define UZ_PL_CIG 256   // staged CIGAR words per span
#define UZ_PL_ESC 64    // staged escape entries per span
// a workgroup barrier that orders LDS traffic only: the round's stores to memory (nobody reads them here) stay in flight across it, where
// __syncthreads() would have every wave wait for their acknowledgement three times a round
#define PL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ __launch_bounds__(256) void k_pack_dict(RecColumns c, int32_t n_tup, uint4 *dict) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_tup) return;
    uint4 e; // flag | l_seq << 16, n_cigar | umask << 16, mapq | aux << 8 | n_low << 16 | n_bl << 24
    e.x = (uint32_t)c.tup_flag[k] | ((uint32_t)c.tup_l_seq[k] << 16);
    e.y = (uint32_t)c.tup_n_cigar[k] | ((uint32_t)c.tup_umask[k] << 16);
    e.z = (uint32_t)c.tup_mapq[k] | ((uint32_t)c.tup_aux[k] << 8) | ((uint32_t)c.tup_n_low[k] << 16) | ((c.tup_n_bl ? (uint32_t)c.tup_n_bl[k] : 0u) << 24);
    e.w = 0u;
    dict[k] = e;
}
__global__ __launch_bounds__(256, 3) void k_pack_link(int64_t n, RecColumns c_in, const unsigned long long *__restrict__ sums, int32_t n_tup, const uint4 *__restrict__ dict, RecA *ra, RecB *rb,
                                                     uint32_t *fm, uint32_t *qoff, uint8_t *nlow, uint16_t *umask_out, uint32_t *plane_out,
                                                     uint16_t *qs, int32_t *coarse, int32_t *mid_idx, int32_t *mid8, int32_t *hflags) {
    RecColumns c = uz_columns_of<true>(c_in);
    __shared__ uint16_t s_tup[UZ_PL_SPAN];
    __shared__ uint8_t s_sd[UZ_PL_SPAN], s_pd[UZ_PL_SPAN];
    __shared__ uint8_t s_qp[UZ_PL_QP], s_blp[UZ_PL_BL], s_blc[UZ_PL_BL / 4 + 8];
    __shared__ uint32_t s_cig[UZ_PL_CIG];
    __shared__ unsigned long long s_esck[UZ_PL_ESC];
    __shared__ int32_t s_escv[UZ_PL_ESC];
    __shared__ RecA s_ra[UZ_PL_SPAN]; // the span's headers: finished here (a SECOND record writes its FIRST's template length), stored once at the end
    __shared__ RecB s_rb[UZ_PL_SPAN];
    __shared__ uint8_t s_back[UZ_PL_SPAN];        // how far back a SECOND's FIRST stands (0: not named inside the span)
    __shared__ uint32_t s_named[UZ_PL_SPAN / 4];  // how often a record was named (a byte each)
    __shared__ uint32_t wsum[UZ_PK_SCANNED][4];
    __shared__ unsigned long long s_run[UZ_PK_SUMS], s_nxt[UZ_PK_SUMS];
    __shared__ int64_t esc_span[2];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
#ifdef UZ_PACK_TIMING
    unsigned long long ptk__[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptl__ = __builtin_amdgcn_s_memtime();
#endif
    const int64_t s0 = (int64_t)blockIdx.x * UZ_PL_SPAN;
    const int cnt = (int)(n - s0 < UZ_PL_SPAN ? n - s0 : UZ_PL_SPAN);
    if (t < UZ_PK_SUMS) { s_run[t] = sums[UZ_PK_SUMS * (size_t)blockIdx.x + t]; s_nxt[t] = sums[UZ_PK_SUMS * ((size_t)blockIdx.x + 1) + t]; }
    if (c.n_esc16 > 0 && t < 2) esc_span[t] = esc_lower_bound(c.esc16_key, c.n_esc16, s0 + (int64_t)t * UZ_PL_SPAN);
    // the byte columns of the span and the dictionary (independent of the sums: requested beside them)
    for (int k = t; k < cnt; k += 256) { s_sd[k] = c.start_d8[s0 + k]; s_pd[k] = c.pair_d8[s0 + k]; }
    for (int k = t; k < UZ_PL_SPAN; k += 256) s_back[k] = 0;
    for (int k = t; k < UZ_PL_SPAN / 4; k += 256) s_named[k] = 0u;
    if (c.tup8 == nullptr) {
        for (int k = t; k < cnt; k += 256) s_tup[k] = c.tup[s0 + k];
        __syncthreads();
    } else {
        // The dictionary index arrives in ONE byte (uz_types.h tup8: its place among the 255 most frequent combinations, or 255 and the 16-bit
        // index in an escape list whose offsets are given per span of 1 024 records -- this workgroup's span).  Rounds 5's k_tup_expand rebuilt the
        // 16-bit column in HBM in front of this kernel (1 byte read + 2 written per record, and this kernel read the 2 again: 0.35 ms of a staged
        // step's chip time, a launch per table); the span's indices are now rebuilt where they are used, in LDS: four records per lane, an
        // escaped record's place = the span's offset + the escapes in front of it (a block scan of the lanes' counts).  The same checks: a span
        // whose escapes are not the number its two offsets name, an escape beyond the list, an index beyond the dictionary -> hflags[0].
        __shared__ uint16_t s_hot[256];
        __shared__ uint32_t s_tsum[4];
        s_hot[t] = c.tup_hot[t];
        const int64_t base = s0 + 4 * t;
        uint32_t w = 0;
        if (base + 4 <= n) w = *reinterpret_cast<const uint32_t *>(c.tup8 + base);
        else for (int k = 0; k < 4; k++) if (base + k < n) w |= (uint32_t)c.tup8[base + k] << (8 * k);
        uint32_t ne = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) ne += (base + k < n && ((w >> (8 * k)) & 0xFFu) == 255u) ? 1u : 0u;
        const uint32_t inc = wv_incl_scan(ne);
        if (lane == 63) s_tsum[wv] = inc;
        __syncthreads();
        uint32_t before = inc - ne;
        for (int k = 0; k < wv; k++) before += s_tsum[k];
        const uint32_t x0 = c.tup_esc_off[blockIdx.x], x1 = c.tup_esc_off[blockIdx.x + 1];
        if (t == 255 && before + ne != x1 - x0) hflags[0] = 1;
        uint32_t at = x0 + before;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t b = (w >> (8 * k)) & 0xFFu;
            uint32_t val = s_hot[b];
            if (b == 255u && base + k < n) {
                if ((int64_t)at < c.n_tup_esc && at < x1) val = c.tup_esc[at];
                else { val = 0; hflags[0] = 1; }
                at++;
            }
            if (val >= (uint32_t)n_tup) { val = 0; if (base + k < n) hflags[0] = 1; }
            s_tup[4 * t + k] = (uint16_t)val;
        }
        __syncthreads();
    }
    PK_TICK(0); // columns + sums rows
    // the dictionary entry of a round's record is requested a round ahead (the first one here, beside the lists)
    auto dict_of = [&](int k) -> uint4 {
        uint32_t tp = k < cnt ? (uint32_t)s_tup[k] : 0u;
        if ((int)tp >= n_tup) { hflags[0] = 1; tp = 0u; } // (an index beyond the dictionary)
        return dict[tp];
    };
    uint4 de_next = dict_of(t);
    // Everything below counts in 32 bits from the span's own base: the running offsets inside the span (rel), the span's share of every
    // quantity (len: next row - this row; the host has checked that the rows ascend and that the totals fit 32 bits), the bases of the five
    // quantities that become absolute offsets.  All of it uniform: read back through readfirstlane so that it lives in scalar registers.
    auto uni64 = [&](unsigned long long x) -> unsigned long long {
        return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) | (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    };
    uint32_t rel[UZ_PK_SCANNED], len[UZ_PK_SCANNED];
#pragma unroll
    for (int k = 0; k < UZ_PK_SCANNED; k++) { rel[k] = 0u; len[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(s_nxt[k] - s_run[k])); }
    const uint32_t base_cg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_run[0]), base_sq = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_run[2]);
    const uint32_t base_st = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_run[5]), base_qn = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s_run[6]);
    const uint32_t base_sql = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)c.n_seq_link + s_run[7]));
    bool len_ok = true; // (a span's share of anything is far below 2^32; a row pair that says otherwise is refused)
#pragma unroll
    for (int k = 0; k < UZ_PK_SCANNED; k++) if (k != 5 && k != 6) len_ok &= s_nxt[k] - s_run[k] <= 0xFFFFFFFFULL;
    if (!len_ok) hflags[0] = 1;
    // this span's share of the lists, where the packer's sums put it
    const unsigned long long q0 = uni64(s_run[3]), b0 = uni64(s_run[8]), g0 = uni64(s_run[4]);
    const uint32_t qn_ = len[3], bn_ = len[8], gn_ = len[4];
    const bool q_in = qn_ <= UZ_PL_QP, b_in = bn_ <= UZ_PL_BL, g_in = gn_ <= UZ_PL_CIG;
    const int64_t e0 = c.n_esc16 > 0 ? (int64_t)uni64((unsigned long long)esc_span[0]) : 0, e1 = c.n_esc16 > 0 ? (int64_t)uni64((unsigned long long)esc_span[1]) : 0;
    const bool e_in = e1 - e0 <= UZ_PL_ESC;
    const uint32_t b0lo = (uint32_t)(b0 & 3ULL); // (four two-bit codes per byte: the span's first code sits b0lo codes into its first byte)
    if (q_in) for (int k = t; k < (int)qn_; k += 256) s_qp[k] = c.qlow_pos[q0 + k];
    if (b_in) {
        for (int k = t; k < (int)bn_; k += 256) s_blp[k] = c.bl_pos[b0 + k];
        const int ncb = (int)((b0lo + bn_ + 3u) >> 2);
        for (int k = t; k < ncb; k += 256) s_blc[k] = c.bl_code[(b0 >> 2) + k];
    }
    if (g_in) for (int k = t; k < (int)gn_; k += 256) s_cig[k] = c.cigar_staged[g0 + k];
    if (e_in) for (int k = t; k < (int)(e1 - e0); k += 256) { s_esck[k] = c.esc16_key[e0 + k]; s_escv[k] = c.esc16_val[e0 + k]; }
    __syncthreads();
    PK_TICK(1); // lists
    auto esc_of = [&](int64_t i, int col) -> int32_t { // value of (record, column) in the escape list, 0 when missing (caught by the totals / the mate check)
        const unsigned long long key = ((unsigned long long)i << 2) | (unsigned long long)col;
        if (e_in) {
            int lo = 0, hi = (int)(e1 - e0);
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_esck[mid] < key) lo = mid + 1; else hi = mid; }
            return (lo < (int)(e1 - e0) && s_esck[lo] == key) ? s_escv[lo] : 0;
        }
        int64_t lo = e0, hi = e1;
        while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (c.esc16_key[mid] < key) lo = mid + 1; else hi = mid; }
        return (lo < c.n_esc16 && c.esc16_key[lo] == key) ? c.esc16_val[lo] : 0;
    };
    // (r: place in the span's share of the list)
    auto qpos_at = [&](uint32_t r) -> int { return q_in ? (int)s_qp[r] : (int)c.qlow_pos[q0 + r]; };
    auto blpos_at = [&](uint32_t r) -> int { return b_in ? (int)s_blp[r] : (int)c.bl_pos[b0 + r]; };
    auto blcode_at = [&](uint32_t r) -> uint32_t {
        const uint32_t g = b0lo + r;
        const uint32_t byte = b_in ? (uint32_t)s_blc[g >> 2] : (uint32_t)c.bl_code[(b0 >> 2) + (g >> 2)];
        return (byte >> (2u * (g & 3u))) & 3u;
    };
    auto cig_at = [&](uint32_t r) -> uint32_t { return g_in ? s_cig[r] : c.cigar_staged[g0 + r]; };
    for (int it = 0; it < UZ_PL_SPAN / 256; it++) {
        const int k = it * 256 + t; // the record's place in the span
        const int64_t i = s0 + k;
        const bool in = k < cnt;
        uint4 de = make_uint4(0u, 0u, 0u, 0u);
        uint32_t p = UZ_P8_NEW, sd = 0;
        if (in) { de = de_next; p = s_pd[k]; sd = s_sd[k]; }
        if (it + 1 < UZ_PL_SPAN / 256) de_next = dict_of(k + 256);
        const uint32_t flag = de.x & 0xFFFFu, ls = de.x >> 16, nc = de.y & 0xFFFFu, um = in ? (de.y >> 16) : UZ_UMASK_ALL;
        const uint32_t mapq = de.z & 0xFFu, ax = (de.z >> 8) & 0xFFu, nb = de.z >> 24;
        const int nl = (int)((de.z >> 16) & 0xFFu);
        uint32_t v[UZ_PK_SUMS], inc[UZ_PK_SCANNED];
        pk_vals(nc, ls, ax, nl, um, nb, v);
        if (!in) {
#pragma unroll
            for (int q = 0; q < UZ_PK_SUMS; q++) v[q] = 0u;
        } else {
            v[5] = sd == UZ_D8_ESC ? (uint32_t)esc_of(i, 0) : sd;
            v[6] = (p != UZ_P8_SECOND && p != UZ_P8_SECOND_TLEN && p != UZ_P8_OLD) ? 1u : 0u;
        }
        PK_TICK(2); // dictionary entry + values
#pragma unroll
        for (int q = 0; q < UZ_PK_SCANNED; q++) inc[q] = wv_incl_scan(v[q]);
        PL_BARRIER(); // wsum of the previous round has been read
        if (lane == 63) {
#pragma unroll
            for (int q = 0; q < UZ_PK_SCANNED; q++) wsum[q][wv] = inc[q];
        }
        PL_BARRIER();
        // where the record's parts end, counted from the span's base: the rounds before, the waves before, the lanes before and the record
        // itself; the running offsets move on at once, so that only the eight offsets below live through the rest of the round.  Nothing
        // of a record may end beyond the span's share: nothing is read or written outside it.
        bool fits = in;
        uint32_t e32[UZ_PK_SCANNED];
#pragma unroll
        for (int q = 0; q < UZ_PK_SCANNED; q++) {
            uint32_t pre = 0, tot = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) { if (w < wv) pre += wsum[q][w]; tot += wsum[q][w]; }
            e32[q] = rel[q] + pre + inc[q];
            if (q != 5 && q != 6) fits &= e32[q] <= len[q] && e32[q] >= inc[q]; // (no wrap)
            rel[q] += tot;
        }
        if (in && !fits) hflags[0] = 1;
        const uint32_t o_cg = base_cg + e32[0] - v[0], o_sq = base_sq + e32[2] - v[2], o_sql = base_sql + e32[7] - v[7];
        const uint32_t r_qp = e32[3] - v[3], r_gw = e32[4] - v[4], r_bl = e32[8] - v[8];
        const uint32_t o_st = base_st + e32[5], o_qn = base_qn + e32[6];
        const uint32_t v3 = v[3];
        PK_TICK(3); // scans
        // ---- first half: everything of the record that does not depend on its mate
        int32_t st0 = 0, en0 = 0, tl0 = 0, mt0 = -1;
        uint32_t qn0 = 0, sq = 0, cg = 0;
        int cig_nonmatch = 0, cig_none = 0;
        if (fits) {
            sq = (ax & UZ_AUX_NO_SEQ) ? UZ_NO_SEQ_OFF : (nb ? o_sql : o_sq);
            cg = o_cg;
            st0 = (int32_t)o_st;
            qn0 = o_qn - 1u; // a new name's id = new names before it
            if (p == UZ_P8_SECOND) { mt0 = -2; qn0 = 0u; }
            else if (p == UZ_P8_SECOND_TLEN) { mt0 = -2; qn0 = 0u; tl0 = esc_of(i, 1); }
            else if (p <= UZ_P8_MAX_DIST) {
                mt0 = (int32_t)(i + p);
                if (mt0 >= n) { hflags[0] = 7; mt0 = -1; }
            } else {
                tl0 = esc_of(i, 1);
                mt0 = esc_of(i, 2);
                if (p == UZ_P8_OLD) qn0 = (uint32_t)esc_of(i, 3);
                if (mt0 < -1 || mt0 >= n) { hflags[0] = 7; mt0 = -1; }
            }
            // CIGAR: the word a simple record's aux byte names, or the words that travelled; `end` and the two counts of the QC word from them
            const uint32_t code = (ax & UZ_AUX_SIMPLE_MASK) >> UZ_AUX_SIMPLE_SHIFT;
            if (code && nc != 1) hflags[0] = 6;
            const uint32_t gw = r_gw;
            int64_t ref_len = 0;
            for (uint32_t q = 0; q < nc; q++) {
                const uint32_t w = code ? uz_cigar_simple_word(code, ls) : cig_at(gw + q);
                const uint32_t op = w & 15u;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += (int64_t)(w >> 4); // M D N = X
                if (!(ax & UZ_AUX_DECODE_BAD)) uz_cigar_op_counts(w, cig_nonmatch, cig_none);
                c.cigar_out[cg + q] = w;
            }
            // (as htslib's bam_endpos -- uz_bam_endpos in pack.hpp states it for the host)
            en0 = ((flag & 4u) || nc == 0) ? st0 + 1 : (int32_t)(st0 + (ref_len > 0 ? ref_len : 1));
            {
                RecA A;
                RecB B;
                uz_pack_rec(A, B, st0, en0, cg, sq, mt0, qn0, (uint16_t)ls, (uint16_t)nc, tl0);
                s_ra[k] = A; s_rb[k] = B;
            }
            if (p >= 1u && p <= UZ_P8_MAX_DIST && mt0 >= 0 && k + (int)p < UZ_PL_SPAN) { // a FIRST whose SECOND lies in this span: tell it
                const int j = k + (int)p;
                s_back[j] = (uint8_t)p;
                if ((atomicAdd(&s_named[j >> 2], 1u << (8 * (j & 3))) >> (8 * (j & 3))) & 0xFFu) hflags[0] = 8; // named twice
            }
        }
        PL_BARRIER();
        PK_TICK(4); // first half (wide columns, CIGAR)
        // ---- second half: the pair.  A FIRST and its SECOND share name id and template length (+-): tlen = max(end, mate's end) - start of
        // the FIRST, or the SECOND's own (UZ_P8_SECOND_TLEN).  The SECOND finishes both headers (its FIRST stands in front of it: written, and
        // behind a barrier).
        if (fits) {
            const uint32_t named = (s_named[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            if (p == UZ_P8_SECOND || p == UZ_P8_SECOND_TLEN) {
                const int d = (int)s_back[k];
                if (d) { // its FIRST stands d records back, in this span
                    const int f = k - d;
                    const RecA FA = s_ra[f];
                    int32_t tlf; // the FIRST's template length
                    if (p == UZ_P8_SECOND_TLEN) tlf = -tl0;
                    else { tlf = (FA.end > en0 ? FA.end : en0) - FA.start; tl0 = -tlf; }
                    s_rb[f].tlen = tlf;
                    s_rb[k].mate = (int32_t)(s0 + f);
                    s_rb[k].qname = s_rb[f].qname;
                    s_rb[k].tlen = tl0;
                }
                // (else: its FIRST lies in the span before this one -- k_pair_link -- or nowhere: the totals of the packer's sums tell)
            } else if (named) hflags[0] = 8; // named as a mate, but not a SECOND record
            // (a FIRST whose SECOND lies in the next span: k_pair_link)
            if ((i & 7) == 0) { // the search index: start of every 4096th record (coarse), of every 64th (mid) and of every 8th (mid8)
                mid8[i >> 3] = st0;
                if ((i & 63) == 0) {
                    mid_idx[i >> 6] = st0;
                    if ((i & 4095) == 0) coarse[i >> 12] = st0;
                }
            }
            fm[i] = uz_pack_fm(flag, mapq, ax);
            const int units = (int)UZ_ROW_UNITS(ls);
            umask_out[i] = (uint16_t)(nb ? (um | UZ_UMASK_LISTED) : um);
            if (um != UZ_UMASK_ALL && (units > 15 || (um >> units) != 0u)) hflags[0] = 5; // a unit beyond the read, or a read too long for a mask
            PK_TICK(5); // pair + header stores
            if (nb) {
                // The listed bases -> the record's rows (one 16-byte row per unit of its mask; bases that were not listed stay code 0, which
                // the readers of a listed record refuse: phase_body.hpp uz_base).  Positions ascend, so the units come in row order.
                if (um == UZ_UMASK_ALL || (ax & UZ_AUX_NO_SEQ)) hflags[0] = 9;
                else {
                    const uint32_t at = r_bl;
                    uint32_t w4r[4] = {0u, 0u, 0u, 0u};
                    int cur = -1, last = -1;
                    uint32_t row = sq, seen = 0u;
                    bool bad = false;
                    for (uint32_t e = 0; e < nb; e++) {
                        const int pos = blpos_at(at + e);
                        const uint32_t code = blcode_at(at + e);
                        const int u = pos >> 5;
                        bad |= pos <= last || pos >= (int)ls || u > 14 || !((um >> u) & 1u);
                        last = pos;
                        if (u != cur) {
                            if (cur >= 0) {
                                reinterpret_cast<uint4 *>(c.seq4_out)[row] = make_uint4(w4r[0], w4r[1], w4r[2], w4r[3]);
                                w4r[0] = w4r[1] = w4r[2] = w4r[3] = 0u;
                                row++;
                            }
                            cur = u;
                            seen |= 1u << (u & 15);
                        }
                        const int k32 = pos & 31; // byte k32 >> 1 of the row, high nibble when k32 is even
                        w4r[k32 >> 3] |= (1u << code) << (8 * ((k32 >> 1) & 3) + ((k32 & 1) ? 0 : 4));
                    }
                    if (cur >= 0) reinterpret_cast<uint4 *>(c.seq4_out)[row] = make_uint4(w4r[0], w4r[1], w4r[2], w4r[3]);
                    if (bad || seen != um) hflags[0] = 9; // a position outside the mask / not ascending, or a unit of the mask without a listed base
                }
            }
            PK_TICK(6); // listed bases
            // list form of the staged plane: the count as it is; a quality row (at the record's base-row position) only for a record whose
            // bits can be asked for, written here from its listed positions
            nlow[i] = (uint8_t)nl;
            const bool listed = v3 == (uint32_t)nl && !(ax & UZ_AUX_NO_SEQ) && nl <= UZ_QLOW_LIST_MAX;
            qoff[i] = listed ? sq : UZ_NO_QLOW_OFF;
            if (listed) {
                const uint32_t at = r_qp;
                bool bad = false;
                int prev = -1;
                for (int e = 0; e < nl; e++) { const int ps = qpos_at(at + e); bad |= ps >= (int)ls || ps <= prev; prev = ps; }
                if (bad) hflags[0] = 4;
                uint32_t at_row = sq; // the quality row holds the same units as the base row: the staged ones
                auto row_word = [&](int u) { // the bits of unit u
                    uint32_t w = 0;
                    for (int e = 0; e < nl; e++) { const int ps = qpos_at(at + e); if ((ps >> 5) == u) w |= 1u << (ps & 31); }
                    return w;
                };
                if (um != UZ_UMASK_ALL) { // one round per staged unit (usually one or two of a read's five)
                    uint32_t m = um;
                    while (m) { const int u = __ffs((int)m) - 1; m &= m - 1u; plane_out[(size_t)at_row++] = row_word(u); }
                } else
                    for (int u = 0; u < units; u++) plane_out[(size_t)at_row++] = row_word(u);
            }
            qs[i] = uz_qs_word(flag, ax, mapq, nl, (int)nc, cig_nonmatch, cig_none);
            PK_TICK(7); // quality list
        }
    }
#ifdef UZ_PACK_TIMING
    if (threadIdx.x == 0) for (int q = 0; q < 10; q++) atomicAdd(&uz_pack_ticks[q], ptk__[q]);
#endif
    PL_BARRIER();
    for (int q = t; q < cnt; q += 256) { ra[s0 + q] = s_ra[q]; rb[s0 + q] = s_rb[q]; }
    if (t == 0) { // the span's own sums must add up to exactly the packer's next row (the two difference columns modulo 2^32)
        bool same = true;
#pragma unroll
        for (int q = 0; q < UZ_PK_SCANNED; q++) same &= rel[q] == len[q];
        if (!same) hflags[0] = 1;
    }
}

// pair form, across spans: a FIRST among the last 256 records of a span of the header build whose SECOND lies in the next span (the pairs inside a
// span were joined by k_pack_rec's own workgroups; a mate lies at most UZ_P8_MAX_DIST = 252 records on)
__global__ __launch_bounds__(256) void k_pair_link(int64_t n, int pk_shift, const uint8_t *__restrict__ pair, const RecA *__restrict__ ra, RecB *rb, int32_t *hflags) {
    const int64_t s1 = ((int64_t)blockIdx.x + 1) << pk_shift;
    const int64_t i = s1 - 256 + threadIdx.x;
    if (i < 0 || i >= n) return;
    const uint32_t p = pair[i];
    if (p < 1u || p > UZ_P8_MAX_DIST) return;
    const int64_t j = i + p;
    if (j >= n || j < s1) return; // (beyond the table: flagged by the header build; inside the span: done there)
    uz_pair_link_one(i, j, pair, ra, rb, hflags);
}

// ---- ASCII uploads (uz_reads_upload): rows re-laid in the packed geometry ---------------------------------
__global__ __launch_bounds__(256) void k_pack_ascii(int64_t n, const RecA *__restrict__ ra, const RecB *__restrict__ rb,
                                                    const uint32_t *__restrict__ cigar_in, const uint32_t *__restrict__ cigar_off_in,
                                                    const uint8_t *__restrict__ seq_in, const uint32_t *__restrict__ sq_off16_in,
                                                    uint32_t *cigar_out, uint8_t *seq4_out, uint16_t *qs, int32_t *hflags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const RecA A = ra[i];
    const RecB B = rb[i];
    int nonmatch = 0, none = 0;
    for (int k = 0; k < (int)B.n_cigar; k++) {
        const uint32_t w = cigar_in[(size_t)cigar_off_in[i] + k];
        cigar_out[(size_t)A.cigar_off + k] = w;
        uz_cigar_op_counts(w, nonmatch, none);
    }
    { // the two CIGAR bits of the QC word (the header build ran before the words were here); a record without CIGAR / SEQ / QUAL keeps 0
        const uint32_t w = qs[i];
        if (w & 0xFF0Fu || B.n_cigar) // (uz_qs_word gave 0 for UZ_AUX_DECODE_BAD: such a record has no operations)
            qs[i] = (uint16_t)((w & ~(UZ_QC_NM5 | UZ_QC_NONE5)) | (nonmatch <= 5 ? UZ_QC_NM5 : 0u) | (none <= 5 ? UZ_QC_NONE5 : 0u));
    }
    const uint8_t *src = seq_in + ((size_t)sq_off16_in[i] << 4);
    uint8_t *dst = seq4_out + (size_t)A.sq_off * UZ_SEQ4_UNIT_BYTES;
    const int ls = B.l_seq, nb = (int)UZ_ROW_UNITS(ls) * UZ_SEQ4_UNIT_BYTES;
    bool bad = false;
    for (int b = 0; b < nb; b++) {
        uint32_t hi = 0, lo = 0;
        if (2 * b < ls) { hi = uz_ascii_nt16(src[2 * b]); bad |= hi == 0xFFu; }
        if (2 * b + 1 < ls) { lo = uz_ascii_nt16(src[2 * b + 1]); bad |= lo == 0xFFu; }
        dst[b] = (uint8_t)(((hi & 15u) << 4) | (lo & 15u));
    }
    if (bad) hflags[0] = 2;
}
__global__ __launch_bounds__(256) void k_build_qlow(int64_t n, const RecA *__restrict__ ra, const RecB *__restrict__ rb,
                                                    const uint32_t *__restrict__ qoff, const uint8_t *__restrict__ qual8,
                                                    const uint32_t *__restrict__ qual_off16, int thr, uint8_t *qlow, uint8_t *nlow, uint16_t *qs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int ls = rb[i].l_seq;
    const uint8_t *src = qual8 + ((size_t)qual_off16[i] << 4);
    uint32_t *dst = reinterpret_cast<uint32_t *>(qlow) + (size_t)qoff[i];
    const int units = (int)UZ_ROW_UNITS(ls);
    int low = 0;
    for (int u = 0; u < units; u++) {
        uint32_t w = 0;
        for (int k = 0; k < 32 && 32 * u + k < ls; k++) w |= (uint32_t)((int)src[32 * u + k] < thr) << k;
        dst[u] = w;
        low += __popc(w);
    }
    nlow[i] = (uint8_t)(low > 255 ? 255 : low);
    // the one bit of the QC word that depends on the threshold: goodread's count of low-quality bases (:43-52)
    const uint32_t w = qs[i];
    qs[i] = (uint16_t)((w & ~UZ_QC_GOOD) | (((w & UZ_QC_GOOD_DISC) && low <= 10 && rb[i].n_cigar <= 10) ? UZ_QC_GOOD : 0u));
}

// cohort batches: the headers of one kid's table copied into the merged table with its bases added
__global__ __launch_bounds__(256) void k_concat_rec(int64_t n, const RecA *__restrict__ sa, const RecB *__restrict__ sb, const uint32_t *__restrict__ sfm,
                                                    const uint32_t *__restrict__ sqo, const uint8_t *__restrict__ snl,
                                                    const uint16_t *__restrict__ sum_, const uint16_t *__restrict__ sqs, RecA *da, RecB *db, uint32_t *dfm, uint32_t *dqo,
                                                    uint8_t *dnl, uint16_t *dum, uint16_t *dqs,
                                                    int32_t rec_base,
                                                    uint32_t cigar_base, uint32_t unit_base, uint32_t seq_base, uint32_t qname_base) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    RecA A = sa[i];
    RecB B = sb[i];
    A.cigar_off += cigar_base;
    if (A.sq_off != UZ_NO_SEQ_OFF) A.sq_off += seq_base;
    if (B.mate >= 0) B.mate += rec_base;
    B.qname += qname_base;
    da[i] = A; db[i] = B; dfm[i] = sfm[i]; dqo[i] = sqo[i] == UZ_NO_QLOW_OFF ? UZ_NO_QLOW_OFF : sqo[i] + unit_base; dnl[i] = snl[i]; dum[i] = sum_[i]; dqs[i] = sqs[i];
}

RD make_rd(const ReadsDev &r) {
    RD R;
    R.ra = (const RecA *)r.rec_a; R.rb = (const RecB *)r.rec_b; R.fm = r.fm;
    R.contig_off = r.contig_off; R.max_span = r.max_span; R.n_contigs = r.n_contigs;
    R.cigar = r.cigar; R.seq4 = r.seq4; R.qlow = r.qlow; R.qoff = r.qoff; R.nlow = r.nlow; R.umask = r.umask; R.qs = r.qs; R.min_map_qual = 0; R.coarse = r.coarse; R.mid = r.mid; R.mid8 = r.mid8;
    R.err = nullptr; // set by the launcher of the per-DNM kernel
    return R;
}

int next_pow2(long long v) {
    long long p = 1;
    while (p < v) p <<= 1;
    return (int)p;
}

} // namespace

namespace {
// two-bit base rows of the host link -> the four-bit rows the kernels read: one lane per 8 staged bytes (32 bases),
// a pure streaming map (8 B in, 16 B out); the padding of a record's last unit expands to code 1 and is never read
__global__ __launch_bounds__(256) void k_expand_seq2(const unsigned long long *__restrict__ in, uint4 *__restrict__ out, size_t n_units) {
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < n_units; u += (size_t)gridDim.x * 256) {
        const unsigned long long w = in[u]; // little-endian: byte b of the row = bits 8b..8b+7
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t lo = uz_seq2_expand_byte((uint32_t)(w >> (16 * q)) & 0xFFu), hi = uz_seq2_expand_byte((uint32_t)(w >> (16 * q + 8)) & 0xFFu);
            o[q] = lo | (hi << 16);
        }
        out[u] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// the listed bases (not A/C/G/T): their BAM codes written over the expanded rows; every entry owns its nibble
__global__ __launch_bounds__(256) void k_patch_exc(int64_t n_exc, const uint32_t *__restrict__ rec, const uint16_t *__restrict__ pos,
                                                   const uint8_t *__restrict__ code, const RecA *__restrict__ ra, const RecB *__restrict__ rb,
                                                   const uint16_t *__restrict__ umask, int64_t n, uint32_t *seq4_words, int32_t *hflags) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_exc) return;
    const uint32_t i = rec[e];
    const uint32_t k = pos[e];
    if ((int64_t)i >= n || k >= rb[i].l_seq || ra[i].sq_off == UZ_NO_SEQ_OFF || code[e] > 15) { hflags[0] = 3; return; }
    uint32_t unit = k >> 5;
    const uint32_t um = umask[i];
    if (um != UZ_UMASK_ALL) {
        if (unit > 14u || !((um >> unit) & 1u)) return; // the unit stayed home: nothing to patch
        unit = (uint32_t)__popc(um & ((1u << unit) - 1u));
    }
    const size_t byte = ((size_t)ra[i].sq_off + unit) * UZ_SEQ4_UNIT_BYTES + ((k & 31u) >> 1);
    const int sh = (int)(8 * (byte & 3)) + ((k & 1) ? 0 : 4);
    uint32_t *wd = seq4_words + (byte >> 2);
    atomicAnd(wd, ~(15u << sh));
    atomicOr(wd, (uint32_t)code[e] << sh);
}
} // namespace

// (sized for the short spans whatever the table gets: 80 bytes per 1024 records)
size_t uz_rec_scratch_bytes(int64_t n) { return (size_t)(((n + (1 << UZ_PK_SHIFT_SMALL) - 1) >> UZ_PK_SHIFT_SMALL) + 2) * (UZ_PK_SUMS + 1) * sizeof(unsigned long long); }

// The dictionary index in one byte (uz_types.h tup8) back to the 16-bit column the header build reads: a workgroup per span of UZ_TUP8_SPAN
// records, four records per lane (one 4-byte load, one 8-byte store); an escaped record's place in the escape list is the span's offset + the
// escapes in front of it inside the span (a block scan of per-lane counts).  A span whose escapes are not the number its two offsets name, an
// escape beyond the list, an index beyond the dictionary: hflags[0] (UZ_E_RANGE at the table's first use).  1 byte read + 2 written per record.
__global__ __launch_bounds__(256) void k_tup_expand(int64_t n, const uint8_t *__restrict__ tup8, const uint16_t *__restrict__ hot, const uint16_t *__restrict__ esc,
                                                    const uint32_t *__restrict__ esc_off, int64_t n_esc, int32_t n_tup, uint16_t *__restrict__ out, int32_t *hflags) {
    __shared__ uint16_t s_hot[256];
    __shared__ uint32_t s_wsum[4];
    static_assert(UZ_TUP8_SPAN == 4 * 256, "four records per lane");
    const int t = threadIdx.x;
    s_hot[t] = hot[t];
    const int64_t base = (int64_t)blockIdx.x * UZ_TUP8_SPAN + 4 * t;
    uint32_t w = 0;
    if (base + 4 <= n) w = *reinterpret_cast<const uint32_t *>(tup8 + base);
    else for (int k = 0; k < 4; k++) if (base + k < n) w |= (uint32_t)tup8[base + k] << (8 * k);
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) cnt += (base + k < n && ((w >> (8 * k)) & 0xFFu) == 255u) ? 1u : 0u;
    // exclusive scan of cnt over the workgroup: within the wave by DPP-free shuffles, across the four waves through LDS
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += v; }
    if ((t & 63) == 63) s_wsum[t >> 6] = inc;
    __syncthreads();
    uint32_t before = inc - cnt;
    for (int k = 0; k < (t >> 6); k++) before += s_wsum[k];
    const uint32_t e0 = esc_off[blockIdx.x], e1 = esc_off[blockIdx.x + 1];
    if (t == 255 && before + cnt != e1 - e0) hflags[0] = 1;
    uint32_t at = e0 + before;
    uint16_t v4[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t b = (w >> (8 * k)) & 0xFFu;
        uint32_t val = s_hot[b];
        if (b == 255u && base + k < n) {
            if ((int64_t)at < n_esc && at < e1) val = esc[at];
            else { val = 0; hflags[0] = 1; }
            at++;
        }
        if (val >= (uint32_t)n_tup) { val = 0; if (base + k < n) hflags[0] = 1; }
        v4[k] = (uint16_t)val;
    }
    if (base + 4 <= n) *reinterpret_cast<uint2 *>(out + base) = make_uint2((uint32_t)v4[0] | ((uint32_t)v4[1] << 16), (uint32_t)v4[2] | ((uint32_t)v4[3] << 16));
    else for (int k = 0; k < 4; k++) if (base + k < n) out[base + k] = v4[k];
}

void uz_build_records(uz_ctx *c, hipStream_t st, ReadsDev &r, const RecColumns &col_in, void *off_scratch) {
    static_assert(sizeof(RecA) == 16 && sizeof(RecB) == 16, "record headers are two 16-byte words");
    if (r.n <= 0) return;
    RecColumns col = col_in;
    if (col.tup8) col.tup = col.tup_out; // the one-byte dictionary index: the 16-bit column is rebuilt below, in HBM (k_tup_expand) or in k_pack_link's LDS
    col.pk_shift = uz_pk_shift(r.n);
    const unsigned nb = (unsigned)((r.n + (1 << col.pk_shift) - 1) >> col.pk_shift);
    unsigned long long *sums = (unsigned long long *)off_scratch;
    // (the escape list is cut at the spans of the passes below by the workgroups themselves; the coarse search index is written by the second
    // pass: two launches gone from every table)
    PkWant want;
    want.cigar = (unsigned long long)r.n_cigar_total; want.units = (unsigned long long)r.n_row_units; want.seq = (unsigned long long)(r.n_seq_units - r.n_bl_units);
    want.qpos = (unsigned long long)r.n_qlow_pos; want.staged = col.cigar_out ? (unsigned long long)r.n_cigar_staged : ~0ULL;
    want.bl_units = (unsigned long long)r.n_bl_units; want.bl = (unsigned long long)r.n_bl;
    static const bool no_link_build = getenv("UZ_BUILD_GENERIC") != nullptr; // (development aid: the general build of the two kernels for every table)
    const bool link_form = uz_link_form(col) && col.cigar_in != nullptr && !no_link_build;
    // the span sums: from the packer (uz_types.h pk_sums: the two kernels below are not launched, k_pack_rec packs from the packer's offsets and holds
    // every span against them), or counted here
    static const bool no_host_sums = getenv("UZ_BUILD_OWN_SUMS") != nullptr; // (development aid: ignore the packer's)
    const bool host_sums = col.pk_sums != nullptr && !no_host_sums;
    static const bool no_lds_build = getenv("UZ_BUILD_FROM_MEMORY") != nullptr; // (development aid: k_pack_rec for every table)
    // (with host sums the scratch of the span sums is free: it holds the packed dictionary)
    const bool from_lds = link_form && host_sums && col.pk_shift == UZ_PK_SHIFT_SMALL && col.n_tup >= 1 && (size_t)col.n_tup * sizeof(uint4) <= uz_rec_scratch_bytes(r.n) &&
                          !col.qpos_wide && !col.bl_wide && (col.tup_n_bl == nullptr || col.bl_pos != nullptr) && !no_lds_build;
    static const bool expand_in_hbm = getenv("UZ_TUP8_IN_HBM") != nullptr; // (development aid: round 5's k_tup_expand in front of k_pack_link too)
    const bool fold_tup8 = col.tup8 != nullptr && from_lds && !expand_in_hbm && (1 << col.pk_shift) == UZ_TUP8_SPAN;
    if (col.tup8 && !fold_tup8) { // every other build reads the 16-bit column from memory: rebuilt there first
        hipLaunchKernelGGL(k_tup_expand, dim3((unsigned)((r.n + UZ_TUP8_SPAN - 1) / UZ_TUP8_SPAN)), dim3(256), 0, st, (int64_t)r.n, col.tup8, col.tup_hot, col.tup_esc,
                           col.tup_esc_off, col.n_tup_esc, (int32_t)col.n_tup, col.tup_out, c->hflags);
        col.tup8 = nullptr; // (the kernels below read col.tup)
    }
    if (host_sums) sums = const_cast<unsigned long long *>(col.pk_sums);
    else if (link_form) hipLaunchKernelGGL((k_off_block_sums<true>), dim3(nb), dim3(256), 0, st, (int64_t)r.n, col, sums);
    else hipLaunchKernelGGL((k_off_block_sums<false>), dim3(nb), dim3(256), 0, st, (int64_t)r.n, col, sums);
    // a table of up to 4096 spans (a chunk of a staged pass: ~2 k): no scan kernel, every workgroup of the second pass adds up the sums in front
    // of its own block; a larger table (the resident 187 M-record one: 46 k spans) gets the one-workgroup scan
    static const bool no_self = getenv("UZ_BUILD_SCAN_KERNEL") != nullptr; // (development aid)
    const bool self = nb <= 4096 && !no_self && !host_sums;
    if (!self && !host_sums) hipLaunchKernelGGL(k_off_scan_sums, dim3(1), dim3(256), 0, st, (int64_t)nb, sums, want, c->hflags);
#define UZ_PACK_LAUNCH(L, S)                                                                                                                              \
    hipLaunchKernelGGL((k_pack_rec<L, S>), dim3(nb), dim3(256), 0, st, (int64_t)r.n, col, (const unsigned long long *)sums, want, (RecA *)r.rec_a, \
                       (RecB *)r.rec_b, r.fm, r.qoff, r.nlow, r.umask, reinterpret_cast<uint32_t *>(r.qlow), r.qs, r.coarse, r.mid, r.mid8, c->hflags, host_sums ? 1 : 0)
    static const bool build_log = getenv("UZ_BUILD_LOG") != nullptr; // (development aid)
    if (build_log)
        fprintf(stderr, "[uz_build_records] n %lld link_form %d host_sums %d pk_shift %d n_tup %lld qpos_wide %d bl_wide %d -> from_lds %d\n", (long long)r.n, (int)link_form,
                (int)host_sums, (int)col.pk_shift, (long long)col.n_tup, (int)col.qpos_wide, (int)col.bl_wide, (int)from_lds);
    if (from_lds) {
        hipLaunchKernelGGL(k_pack_dict, dim3((unsigned)((col.n_tup + 255) / 256)), dim3(256), 0, st, col, (int32_t)col.n_tup, (uint4 *)off_scratch);
        hipLaunchKernelGGL(k_pack_link, dim3(nb), dim3(256), 0, st, (int64_t)r.n, col, (const unsigned long long *)sums, (int32_t)col.n_tup, (const uint4 *)off_scratch, (RecA *)r.rec_a,
                           (RecB *)r.rec_b, r.fm, r.qoff, r.nlow, r.umask, reinterpret_cast<uint32_t *>(r.qlow), r.qs, r.coarse, r.mid, r.mid8, c->hflags);
    } else if (link_form) { if (self) UZ_PACK_LAUNCH(true, true); else UZ_PACK_LAUNCH(true, false); }
    else { if (self) UZ_PACK_LAUNCH(false, true); else UZ_PACK_LAUNCH(false, false); }
#undef UZ_PACK_LAUNCH
#ifdef UZ_PACK_TIMING
    {
        UZ_HIP(hipStreamSynchronize(st));
        unsigned long long tk[16];
        UZ_HIP(hipMemcpyFromSymbol(tk, HIP_SYMBOL(uz_pack_ticks), sizeof(tk)));
        const char *nm[10] = {"s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9"};
        unsigned long long tot = 0;
        for (int k = 0; k < 10; k++) tot += tk[k];
        fprintf(stderr, "[pack timing] n %lld wgs %u", (long long)r.n, nb);
        for (int k = 0; k < 10; k++) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)tk[k] / (double)(tot ? tot : 1));
        fprintf(stderr, " | ticks/wg %.0f\n", (double)tot / nb);
        unsigned long long z[16] = {0};
        UZ_HIP(hipMemcpyToSymbol(HIP_SYMBOL(uz_pack_ticks), z, sizeof(z)));
    }
#endif
    if (col.pair_d8)
        hipLaunchKernelGGL(k_pair_link, dim3(nb), dim3(256), 0, st, (int64_t)r.n, (int)col.pk_shift, col.pair_d8, (const RecA *)r.rec_a, (RecB *)r.rec_b,
                           c->hflags);
    // the table arrived with two-bit base rows: expand them into seq4 (the units of list-form records, behind them, were written by
    // k_pack_rec); then the listed bases that are not A/C/G/T, of either kind of record
    if (r.seq2_staged && r.n_seq_units - r.n_bl_units > 0) {
        const size_t nu = (size_t)(r.n_seq_units - r.n_bl_units);
        hipLaunchKernelGGL(k_expand_seq2, dim3((unsigned)std::min<size_t>((nu + 255) / 256, 16384)), dim3(256), 0, st,
                           (const unsigned long long *)r.seq2_staged, (uint4 *)const_cast<uint8_t *>(r.seq4), nu);
    }
    if (r.n_exc > 0 && r.exc_rec) {
        hipLaunchKernelGGL(k_patch_exc, dim3((unsigned)((r.n_exc + 255) / 256)), dim3(256), 0, st, r.n_exc, r.exc_rec, r.exc_pos, r.exc_code,
                           (const RecA *)r.rec_a, (const RecB *)r.rec_b, (const uint16_t *)r.umask, (int64_t)r.n,
                           (uint32_t *)const_cast<uint8_t *>(r.seq4), c->hflags);
        r.n_exc = 0;
    }
    r.seq2_staged = nullptr;
    UZ_HIP(hipGetLastError());
}

void uz_concat_table(uz_ctx *c, hipStream_t st, ReadsDev &dst, const ReadsDev &src, int64_t rec_base, int64_t cigar_base, int64_t unit_base,
                     int64_t seq_base, uint32_t qname_base) {
    if (src.n <= 0) return;
    hipLaunchKernelGGL(k_concat_rec, dim3((unsigned)((src.n + 255) / 256)), dim3(256), 0, st, (int64_t)src.n, (const RecA *)src.rec_a,
                       (const RecB *)src.rec_b, (const uint32_t *)src.fm, (const uint32_t *)src.qoff, (const uint8_t *)src.nlow,
                       (const uint16_t *)src.umask, (const uint16_t *)src.qs, (RecA *)dst.rec_a + rec_base, (RecB *)dst.rec_b + rec_base, dst.fm + rec_base,
                       dst.qoff + rec_base, dst.nlow + rec_base, dst.umask + rec_base, dst.qs + rec_base,
                       (int32_t)rec_base, (uint32_t)cigar_base,
                       (uint32_t)unit_base, (uint32_t)seq_base, qname_base);
    UZ_HIP(hipGetLastError());
    if (src.n_cigar_total)
        UZ_HIP(hipMemcpyAsync(const_cast<uint32_t *>(dst.cigar) + cigar_base, src.cigar, (size_t)src.n_cigar_total * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    if (src.n_seq_units)
        UZ_HIP(hipMemcpyAsync(const_cast<uint8_t *>(dst.seq4) + (size_t)seq_base * UZ_SEQ4_UNIT_BYTES, src.seq4, (size_t)src.n_seq_units * UZ_SEQ4_UNIT_BYTES,
                              hipMemcpyDeviceToDevice, st));
    if (src.n_plane_units) {
        UZ_HIP(hipMemcpyAsync(dst.qlow + (size_t)unit_base * UZ_QLOW_UNIT_BYTES, src.qlow, (size_t)src.n_plane_units * UZ_QLOW_UNIT_BYTES,
                              hipMemcpyDeviceToDevice, st));
    }
}

void uz_finish_table(uz_ctx *c, hipStream_t st, ReadsDev &r) {
    if (r.n <= 0) return;
    const int64_t nk = (r.n >> 3) + 2;
    hipLaunchKernelGGL(k_build_coarse, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, (const RecA *)r.rec_a, (int64_t)r.n, r.coarse, r.mid, r.mid8);
    UZ_HIP(hipGetLastError());
}

void uz_pack_ascii_rows(uz_ctx *c, hipStream_t st, ReadsDev &r, const uint32_t *cigar_in, const uint32_t *cigar_off_in,
                        const uint8_t *seq_in, const uint32_t *sq_off16_in, uint32_t *cigar_out, uint8_t *seq4_out) {
    if (r.n <= 0) return;
    hipLaunchKernelGGL(k_pack_ascii, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, st, (int64_t)r.n, (const RecA *)r.rec_a,
                       (const RecB *)r.rec_b, cigar_in, cigar_off_in, seq_in, sq_off16_in, cigar_out, seq4_out, r.qs, c->hflags);
    UZ_HIP(hipGetLastError());
}

void uz_build_qlow(uz_ctx *c, hipStream_t st, ReadsDev &r, int min_base_qual) {
    const int thr = min_base_qual < 0 ? 0 : (min_base_qual > 255 ? 256 : min_base_qual);
    if (r.n > 0) {
        hipLaunchKernelGGL(k_build_qlow, dim3((unsigned)((r.n + 255) / 256)), dim3(256), 0, st, (int64_t)r.n, (const RecA *)r.rec_a,
                           (const RecB *)r.rec_b, (const uint32_t *)r.qoff, (const uint8_t *)r.qual8, (const uint32_t *)r.qual_off16, thr, r.qlow, r.nlow, r.qs);
        UZ_HIP(hipGetLastError());
    }
    r.qlow_thr = min_base_qual;
    r.qlow_valid = true;
}

void uz_phase_state_free(uz_ctx *c) {
    PhaseState *st = (PhaseState *)c->phase_state;
    if (!st) return;
    st->scratch.release(); st->bounds.release(); st->res.release();
    st->cursor.release(); st->pre_win.release(); st->pre_h.release(); st->pool.release(); st->list_len.release(); st->list_start.release();
    st->pool_cursor = nullptr; st->need_count.release();
    if (st->bounds_ready) (void)hipEventDestroy(st->bounds_ready);
    if (st->bounds_h) (void)hipHostFree(st->bounds_h);
    delete st;
    c->phase_state = nullptr;
}

int uz_want_lists = 1; // uz_phase always keeps the lists available for uz_phase_votes / uz_phase_groups

static void phase_check_upload_flags(uz_ctx *c) {
    if (c->hflags[0]) { // set by the header build of an upload (abi.hip) whose commands have now run
        const int f = c->hflags[0];
        c->hflags[0] = 0;
        throw UzError{UZ_E_RANGE, f == 2 ? "SEQ holds a character outside BAM's 16-code alphabet"
                                  : f == 3 ? "exc_* columns of the reads view: an entry names a record without bases, a base beyond l_seq or a code above 15"
                                  : f == 4 ? "qlow_pos of the reads view: positions of a record are not ascending or lie beyond l_seq"
                                  : f == 5 ? "umask of the reads view: a unit beyond the read's length, or a mask on a read longer than 480 bases"
                                  : f == 6 ? "aux of the reads view: a simple-CIGAR code on a record whose n_cigar is not 1"
                                  : f == 7 ? "mate_d / esc16_* of the reads view: a mate index outside the table"
                                  : f == 8 ? "pair_d8 of the reads view: a SECOND record that no FIRST record names, or one that two of them name"
                                  : f == 9 ? "bl_* of the reads view: a listed base outside the record's unit mask or its length, positions not ascending, a unit of the mask without a listed base, or a list on a record without a mask / without bases"
                                           : "n_cigar_total / n_row_units of the reads view do not match its columns"};
    }
}
struct Sizes { Caps caps; int arena, arena2; long long sumP; };
static const int arena_env = [] { const char *e = getenv("UZ_PHASE_LDS_KB"); return e ? atoi(e) * 1024 : -1; }();
// the share of a batch's DNMs (per mille, by the estimate below) the arena is sized to hold; the rest take the HBM build behind it
// (1000 / 990 / 940 / 900 on the bench batch: 3.16 / 2.98 / 2.84 / 2.84 ms -- one more wave per CU is worth more than the 1.3 % of the DNMs redone)
static const int arena_permille = [] { const char *e = getenv("UZ_PHASE_ARENA_PERMILLE"); return e ? atoi(e) : 940; }();
// What the host needs of a batch's bounds (k_phase_bounds: five numbers per DNM) is a handful of maxima, one sum and a histogram: reduced on
// the device (k_bounds_reduce) behind the sizing pass, 1.1 KB come back instead of 20 bytes per DNM -- through round 5 the host walked the 2 MB
// of a 100 k-DNM batch twice between the device's last kernel and the call's return (0.24 ms of a 4.3 ms resident step with the device idle).
struct BoundsRed {
    int32_t mA, mT, mH, mC, active, pad;
    unsigned long long mM, sumP; // (mM by atomicMax on 64 bits)
    int32_t hist[256];           // arena estimate in units of 256 bytes, DNMs with candidate sites only
};
// (every workgroup ends with ~17 atomics on ONE 1 KB record, which the L2 takes one after the other -- ~5 ns each: 391 workgroups were 37 us for
// a pass over 2 MB; a few dozen workgroups walking more DNMs each are not)
#ifndef UZ_BR_BLOCKS
#define UZ_BR_BLOCKS 48
#endif
__global__ __launch_bounds__(256) void k_bounds_reduce(const int32_t *__restrict__ bounds, int32_t n, BoundsRed *out) {
    __shared__ int32_t hist[256];
    __shared__ unsigned long long part[4][8];
    hist[threadIdx.x] = 0;
    __syncthreads();
    long long mA = 0, mT = 0, mH = 0, mC = 0, mM = 0, sumP = 0, active = 0;
    int bin = -1;
    for (int64_t d0 = (int64_t)blockIdx.x * 256; d0 < n; d0 += (int64_t)gridDim.x * 256) { // (whole wavefronts stay in the loop: the ballots below)
        const int64_t d = d0 + threadIdx.x;
        const bool in = d < n;
        const int32_t *b = bounds + 5 * (in ? d : 0);
        const long long b0 = in ? b[0] : 0, b1 = in ? b[1] : 0, b2 = in ? b[2] : 0, b3 = in ? b[3] : 0, b4 = in ? b[4] : 0;
        mA = max(mA, b0); mT = max(mT, b1); mH = max(mH, b2); mC = max(mC, b3);
        const long long M = b1 + 4LL * b0 * (b4 + 1);
        mM = max(mM, M);
        sumP += min(M, 4096LL) + b3;
        if (b3 > 0) {
            const long long est = ((37LL * b1) / 4 + 10LL * b0 + 3328 + 255) >> 8;
            bin = (int)min(est, 255LL);
            active++;
        }
        // (a batch's estimates fall into a handful of bins: one LDS atomic per distinct bin of the wavefront, not 64 on the same word)
        unsigned long long left = __ballot(bin >= 0);
        while (left) {
            const int bb = __shfl(bin, __ffsll((long long)left) - 1, 64);
            const unsigned long long m = __ballot(bin == bb);
            if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) atomicAdd(&hist[bb], (int32_t)__popcll(m));
            left &= ~m;
        }
        bin = -1;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mA = max(mA, __shfl_xor(mA, o, 64)); mT = max(mT, __shfl_xor(mT, o, 64)); mH = max(mH, __shfl_xor(mH, o, 64)); mC = max(mC, __shfl_xor(mC, o, 64));
        mM = max(mM, __shfl_xor(mM, o, 64)); sumP += __shfl_xor(sumP, o, 64); active += __shfl_xor(active, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[w][0] = (unsigned long long)mA; part[w][1] = (unsigned long long)mT; part[w][2] = (unsigned long long)mH; part[w][3] = (unsigned long long)mC;
        part[w][4] = (unsigned long long)mM; part[w][5] = (unsigned long long)sumP; part[w][6] = (unsigned long long)active;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long r[7];
        for (int k = 0; k < 7; k++) { r[k] = part[0][k]; for (int q = 1; q < 4; q++) r[k] = k < 5 ? max(r[k], part[q][k]) : r[k] + part[q][k]; }
        atomicMax(&out->mA, (int32_t)r[0]); atomicMax(&out->mT, (int32_t)r[1]); atomicMax(&out->mH, (int32_t)r[2]); atomicMax(&out->mC, (int32_t)r[3]);
        atomicMax(&out->mM, r[4]); atomicAdd(&out->sumP, r[5]); atomicAdd(&out->active, (int32_t)r[6]);
    }
    if (hist[threadIdx.x]) atomicAdd(&out->hist[threadIdx.x], hist[threadIdx.x]);
}
static Sizes phase_exact_sizes(const BoundsRed *br, int32_t n) {
    const long long mA = br->mA, mT = br->mT, mH = br->mH, mC = br->mC, mM = (long long)br->mM, sumP = (long long)br->sumP;
    Sizes z;
    z.caps.A = (int32_t)mA; z.caps.T = (int32_t)mT; z.caps.H = (int32_t)mH; z.caps.C = (int32_t)mC;
    z.caps.I = (int32_t)(4 * mA);
    z.caps.M = next_pow2(std::min<long long>(std::max<long long>(mM, 2), 1 << 20));
    z.sumP = sumP;
    // LDS arena of k_phase<true>, sized for THIS batch.  One wave works on a DNM, so the arena alone decides how many DNMs a CU has in
    // flight: a DNM needs about 9.2 bytes per record its het-site fetches return plus 3.2 KiB (fit over the bench workload by the CPU twin,
    // scripts/phase_sizes.py: residual p99 0.4 KiB).  The 94th percentile of that estimate over the batch decides how
    // many waves share a CU's 160 KiB (at most 4 x UZ_PHASE_MIN_WAVES: registers), and the arena is then the largest that this many leave
    // room for.  The bench batch runs 16 DNMs per CU on 9.25 KiB arenas; a deep-coverage batch gets larger arenas and fewer waves instead
    // of handing most of its DNMs to the slower HBM build.
    z.arena = arena_env;
    if (z.arena < 0) {
        // (k_bounds_reduce: est = ((37 b[1]) / 4 + 10 b[0] + 3328 + 255) >> 8 over the DNMs with candidate sites -- b[0]: records of the DNM's own
        // fetch, 33 of a point variant, hundreds around an SV's breakpoints: a class byte and two flags each)
        const int32_t *hist = br->hist;
        const int32_t active = br->active;
        int u = 16, seen = 0, umax = 16;
        for (int k = 0; k < 256; k++) if (hist[k]) umax = k;
        for (int k = 0; k < 256; k++) {
            seen += hist[k];
            if (hist[k]) u = std::max(u, k);
            if ((long long)seen * 1000 >= (long long)active * arena_permille) break;
        }
        z.arena = std::min(u * 256, 62 * 1024);
        z.arena2 = std::min(std::max(2 * z.arena, umax * 256 + 1024), 62 * 1024);
    } else
        z.arena2 = std::min(2 * z.arena, 62 * 1024);
    return z;
}

void uz_launch_phase(uz_ctx *c, FamilyDev &f, const SitesDev &s, ReadsDev &r, int32_t *status, int32_t *counts,
                     int32_t *origin, int32_t *evidence, bool defer) {
    if (!c->phase_state) c->phase_state = new PhaseState();
    PhaseState *st = (PhaseState *)c->phase_state;
    const int32_t n = c->dn.n;
    st->n = n;
    st->have_lists = false;
    st->pending = false;
    c->phase_n = n;
    if (n <= 0) { c->phase_valid = true; return; }
    static const bool host_trace = getenv("UZ_PHASE_HOST_TRACE") != nullptr; // development aid (abi.hip: phase_whole)
    const auto host_t0 = std::chrono::steady_clock::now();
    auto host_us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count(); };
    double t_queued = 0, t_synced = 0, t_sized = 0;
    // an asynchronous upload of this table must have landed before the first kernel reads it
    uz_reads_make_ready(c, r);
    // the quality plane holds ONE threshold: tables that kept their full qualities are re-thresholded, the others
    // were decoded for a threshold and refuse another
    if (!r.qlow_valid || r.qlow_thr != c->P.min_gt_qual) {
        UZ_REQUIRE(r.qual8 != nullptr, UZ_E_STATE,
                   "the reads table was packed for --min-gt-qual " + std::to_string(r.qlow_thr) + ", the parameters say " +
                       std::to_string(c->P.min_gt_qual) + ": decode it again for this threshold");
        uz_build_qlow(c, c->stream, r, c->P.min_gt_qual);
    }

    PhaseArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    a.min_gt_qual = c->P.min_gt_qual; a.readlen = c->P.readlen; a.no_extended = c->P.no_extended;
    a.read_goal = c->P.read_goal; a.evidence_min_ratio = c->P.evidence_min_ratio; a.split_error_margin = c->P.split_error_margin;
    a.cutoff = c->dn.cutoff;
    a.cutoff_d = c->cohort_on ? c->dn_cutoff.p : nullptr;
    a.spos = s.pos; a.sref = s.ref_base; a.salt = s.alt_base;
    a.cand_off = c->cand_off.p; a.het_off = c->het_off.p;
    a.cand_idx = c->cand_idx.p; a.het_idx = c->het_idx.p; a.cand_flags = c->cand_flags.p;
    a.rcontig = c->dn.rcontig.p; a.dstart = c->dn.start.p; a.dend = c->dn.end.p; a.dflags = c->dn.dflags.p; a.vartype = c->dn.vartype.p;
    a.allele_off = c->dn.allele_off.p; a.alleles = c->dn.alleles.p;
    a.R = make_rd(r);
    a.R.err = c->hflags + 1;
    a.R.min_map_qual = c->P.min_map_qual;

    // sizing pass -> scratch capacities (max over the batch)
    st->bounds.ensure((size_t)5 * n);
    st->pre_win.ensure((size_t)4 * n); st->pre_h.ensure(2 * ((size_t)c->n_het + 1));
    a.pre_win = st->pre_win.p; a.pre_ha = st->pre_h.p; a.pre_hl = st->pre_h.p + (size_t)c->n_het + 1;
    // DNMs without candidates leave their het ranges untouched: they must read as empty
    UZ_HIP(hipMemsetAsync(st->pre_h.p, 0, 2 * ((size_t)c->n_het + 1) * sizeof(int32_t), c->stream));
    {
        const unsigned nb = (unsigned)(((int64_t)n * UZ_BW_LANES + 255) / 256);
        ProfScope ps(c, UZ_K_SIZING);
        UZ_TRACE("k_phase_bounds");
        hipLaunchKernelGGL(k_phase_bounds, dim3(nb), dim3(256), 0, c->stream, a, st->bounds.p);
        UZ_HIP(hipGetLastError());
    }
    if (st->bounds_h_cap < sizeof(BoundsRed) / sizeof(int32_t)) {
        if (st->bounds_h) (void)hipHostFree(st->bounds_h);
        st->bounds_h = nullptr;
        st->bounds_h_cap = sizeof(BoundsRed) / sizeof(int32_t) + 64;
        UZ_HIP(hipHostMalloc((void **)&st->bounds_h, st->bounds_h_cap * sizeof(int32_t), hipHostMallocDefault));
    }
    st->bounds_red.ensure(sizeof(BoundsRed) / sizeof(int32_t) + 16);
    UZ_HIP(hipMemsetAsync(st->bounds_red.p, 0, sizeof(BoundsRed), c->stream));
    hipLaunchKernelGGL(k_bounds_reduce, dim3((unsigned)std::max(1, std::min(UZ_BR_BLOCKS, (n + 255) / 256))), dim3(256), 0, c->stream, (const int32_t *)st->bounds.p, n,
                       reinterpret_cast<BoundsRed *>(st->bounds_red.p));
    UZ_HIP(hipGetLastError());
    // The sizes of the batch (k_phase_bounds) decide the scratch capacities, the LDS arena and the grid.  Waiting for them costs a
    // host round trip per batch -- a staged pass makes one per chunk.  So a batch is first run SPECULATIVELY on the capacities of the
    // batch before it (with head room): the kernels refuse what does not fit (UZ_ST_CAPACITY, never a write out of bounds), the
    // bounds come back with the results, and only when they exceed what was assumed is the batch run again on its own sizes.
    // (UZ_PHASE_NO_SPEC=1, the capacity / arena test hooks and the first batch of a context take the exact path.)
    const BoundsRed *const bh_own = reinterpret_cast<const BoundsRed *>(st->bounds_h); // (kept apart from the result staging below: both are read after the run)
    uz_kcopy(c, st->bounds_h, st->bounds_red.p, sizeof(BoundsRed));
    if (!st->bounds_ready) UZ_HIP(hipEventCreateWithFlags(&st->bounds_ready, hipEventDisableTiming));
    UZ_HIP(hipEventRecord(st->bounds_ready, c->stream));
    auto check_upload_flags = [&] { phase_check_upload_flags(c); };
    auto exact_sizes = [&](const BoundsRed *bh) { return phase_exact_sizes(bh, n); };
    if (st->n_cus <= 0) { // asked once: the query is not cheap
        hipDeviceProp_t prop;
        UZ_HIP(hipGetDeviceProperties(&prop, c->device));
        st->n_cus = prop.multiProcessorCount;
    }
    static const bool no_spec = getenv("UZ_PHASE_NO_SPEC") != nullptr;
    const bool hooks = getenv("UZ_TEST_CAP_T") || getenv("UZ_TEST_PHASE_ARENA");
    bool speculative = st->spec_valid && !no_spec && !hooks && !st->force_exact;
    st->force_exact = false;
    const size_t res_ints = (size_t)7 * n + ((7 * (size_t)n) & 1); // the results, padded to eight bytes; the pool's fill level behind them
    st->res.ensure(res_ints + 4);
    st->pool_cursor = (unsigned long long *)(st->res.p + res_ints);
    // the work cursors, and behind them the counts of the two lists of given-up DNMs: ONE block, cleared by one memset per attempt
    st->cursor.ensure(16 * (UZ_PHASE_PARTS + 2) + 32);
    // two lists of given-up DNMs: what the first launch leaves for the second (larger arenas), what the second leaves for the HBM build
    st->retry.ensure(2 * ((size_t)n + 16));
    int32_t *const retry1_list = st->retry.p, *const retry2_list = st->retry.p + (size_t)n + 16;
    int32_t *const retry1 = st->cursor.p + 16 * (UZ_PHASE_PARTS + 2), *const retry2 = retry1 + 16;
    a.retry_count = retry1; a.retry_list = retry1_list;
    a.from_list = 0; a.cursor_slot = 0; a.src_count = nullptr; a.src_list = nullptr;
    a.status = st->res.p; a.counts = st->res.p + n; a.origin = st->res.p + (size_t)5 * n; a.evidence = st->res.p + (size_t)6 * n;
    a.work_cursor = st->cursor.p;
    a.want_lists = uz_want_lists;
    st->list_start.ensure(n); st->list_len.ensure((size_t)6 * n);
    a.pool_cursor = st->pool_cursor;
    a.list_start = st->list_start.p; a.list_len = st->list_len.p;
#ifdef UZ_PHASE_TIMING
    static DevBuf<unsigned long long> timing;
    timing.ensure(32);
    UZ_HIP(hipMemsetAsync(timing.p, 0, 32 * sizeof(unsigned long long), c->stream));
    a.timing = timing.p;
#endif
    // pinned staging of the per-DNM results (+ the list-pool fill level): everything comes back behind ONE sync
    {
        const size_t need = (size_t)7 * n + 16;
        if (st->res_h_cap < need) {
            if (st->res_h) (void)hipHostFree(st->res_h);
            st->res_h = nullptr;
            st->res_h_cap = need + (size_t)n;
            UZ_HIP(hipHostMalloc((void **)&st->res_h, st->res_h_cap * sizeof(int32_t), hipHostMallocDefault));
        }
    }
    int32_t *const hres = st->res_h;
    unsigned long long *const hused = (unsigned long long *)(hres + (size_t)7 * n + ((7 * (size_t)n) & 1)); // 8-byte aligned slot
    for (int round = 0; round < 2; round++) {
        Sizes z;
        if (speculative) {
            z.caps = st->spec_caps; z.arena = st->spec_arena; z.arena2 = st->spec_arena2;
            z.sumP = (long long)(st->spec_sumP_per_dnm * (double)n) + 4096;
        } else {
            UZ_HIP(hipEventSynchronize(st->bounds_ready));
            check_upload_flags();
            z = exact_sizes(bh_own);
            if (const char *e = getenv("UZ_TEST_CAP_T")) { // test hook: a scratch too small for some DNMs -> they must come back as UZ_ST_CAPACITY
                const int32_t t = (int32_t)atoi(e);
                if (t < z.caps.T) z.caps.T = t;
            }
        }
        const Caps caps = z.caps;
        int arena_used = z.arena;
        // two layouts of a workgroup's scratch region: the arena build's (what that build keeps in HBM: little) for the two arena launches,
        // the HBM build's (every array) for the launch behind them; one buffer serves both, the launches run one after the other
        ScrOff so_hbm;
        const size_t per_wg = uz_scratch_layout(caps, a.so, true), per_wg_hbm = uz_scratch_layout(caps, so_hbm, false);
        if (const char *e = getenv("UZ_TEST_PHASE_ARENA")) { // test hook: an arena (bytes) too small for most DNMs -> they take the HBM build of k_phase
            const int t = atoi(e);
            if (t >= 0 && t < arena_used) arena_used = t;
        }
        int wgs_per_cu;
        {
            static const int wgs_env = [] { const char *e = getenv("UZ_PHASE_WGS_PER_CU"); return e ? atoi(e) : 0; }();
            const int by_lds = (160 * 1024) / (arena_used + (int)sizeof(WgSharedT<1>) + 512);
            const int by_regs = UZ_PHASE_MIN_WAVES * 4 / (WG_NT / 64); // waves per SIMD x four SIMDs / waves per workgroup
            wgs_per_cu = wgs_env > 0 ? wgs_env : std::max(1, std::min(by_lds, by_regs));
            if (arena_env < 0 && wgs_env <= 0) { // all the room this occupancy leaves
                const int room = ((160 * 1024) / wgs_per_cu - (int)sizeof(WgSharedT<1>) - 512) & ~255;
                if (room > arena_used && !getenv("UZ_TEST_PHASE_ARENA")) arena_used = std::min(room, 62 * 1024);
            }
        }
        a.lds_arena_bytes = arena_used;
        int grid = st->n_cus * wgs_per_cu;
        if (grid > n) grid = n;
        const size_t budget = (size_t)8 << 30; // keep the scratch under 8 GiB
        while (grid > 1 && (size_t)grid * per_wg > budget) grid /= 2;
        int grid_hbm = std::min(grid, 8 * st->n_cus); // (the DNMs the arena launches gave up: usually a handful -- eight waves per CU pull them from the list)
        while (grid_hbm > 1 && (size_t)grid_hbm * per_wg_hbm > budget) grid_hbm /= 2;
        st->scratch.ensure(std::max((size_t)grid * per_wg, (size_t)grid_hbm * per_wg_hbm));
        a.scratch = st->scratch.p; a.scratch_per_wg = per_wg; a.caps = caps;
        size_t pool_cap = (size_t)std::min<long long>(4 * z.sumP + 1024, (long long)1 << 31);
        st->pool.ensure(pool_cap);
        a.pool = st->pool.p; a.pool_cap = pool_cap;
        for (int attempt = 0; attempt < 4; attempt++) {
            UZ_HIP(hipMemsetAsync(st->cursor.p, 0, (16 * (UZ_PHASE_PARTS + 2) + 32) * sizeof(int32_t), c->stream)); // (cursors + both give-up counts)
            UZ_HIP(hipMemsetAsync(st->pool_cursor, 0, 2 * sizeof(unsigned long long), c->stream));
            {
                // Three launches.  The arena is sized for most of the batch's DNMs, not all (the more DNMs a CU holds, the faster the batch);
                // the rest are redone by a second launch of the same build with arenas for the largest estimate -- a DNM's latency in the
                // arena build is ~100 us, in the HBM build ~450 us, and a list that is worked off behind the main launch costs the batch its
                // slowest DNM -- and only what that launch gives up too (a field width, a chaining level with more than 96 winners) goes to
                // the HBM build.
                ProfScope ps(c, UZ_K_PHASE);
                UZ_TRACE("k_phase");
                hipLaunchKernelGGL((k_phase<true>), dim3((unsigned)grid), dim3(WG_NT), (size_t)a.lds_arena_bytes, c->stream, a);
                UZ_HIP(hipGetLastError());
                PhaseArgs a2 = a;
                a2.from_list = 1; a2.cursor_slot = UZ_PHASE_PARTS; a2.src_count = retry1; a2.src_list = retry1_list;
                a2.retry_count = retry2; a2.retry_list = retry2_list;
                a2.lds_arena_bytes = getenv("UZ_TEST_PHASE_ARENA") ? arena_used /* test hook: the small arena again, so that the HBM build is what runs */
                                                                    : std::max(arena_used, std::min(z.arena2, 62 * 1024));
                const int per_cu2 = std::max(1, (160 * 1024) / (a2.lds_arena_bytes + (int)sizeof(WgSharedT<1>) + 512));
                hipLaunchKernelGGL((k_phase<true>), dim3((unsigned)std::min(grid, per_cu2 * st->n_cus)), dim3(WG_NT), (size_t)a2.lds_arena_bytes, c->stream, a2);
                UZ_HIP(hipGetLastError());
                PhaseArgs a3 = a2;
                a3.cursor_slot = UZ_PHASE_PARTS + 1; a3.src_count = retry2; a3.src_list = retry2_list;
                a3.so = so_hbm; a3.scratch_per_wg = per_wg_hbm;
                hipLaunchKernelGGL((k_phase<false>), dim3((unsigned)grid_hbm), dim3(WG_NT), 0, c->stream, a3);
                UZ_HIP(hipGetLastError());
            }
            UZ_TRACE("after k_phase");
            *hused = 0;
            uz_kcopy(c, hres, st->res.p, res_ints * sizeof(int32_t) + sizeof(unsigned long long)); // the results and the pool's fill level: one copy
            int32_t *const hretry = (int32_t *)(hused + 1);
            *hretry = 0;
            uz_kcopy(c, hretry, retry2, sizeof(int32_t)); // DNMs that took the HBM build
            if (defer && speculative) { // uz_phase_begin: the run stays in flight; uz_finish_phase waits for it and judges it
                st->pending = true;
                st->pend_caps = caps; st->pend_pool_cap = a.pool_cap; st->pend_want_lists = a.want_lists;
                return;
            }
            if (host_trace) t_queued = host_us();
            UZ_HIP(hipStreamSynchronize(c->stream));
            if (host_trace) t_synced = host_us();
            if (speculative) check_upload_flags();
            if (c->hflags[1]) {
                const int f = c->hflags[1];
                c->hflags[1] = 0;
                throw UzError{UZ_E_STATE, f == 4 ? "the read stage asked for a base of a record whose bases were staged as a list (bl_*), and the list does not hold it: the fetch points that staged the table do not cover this batch"
                                        : f == 3 ? "the read stage asked for a base (or its quality bit) in a 32-base unit that was not staged (umask): the fetch points that staged the table do not cover this batch"
                                        : f == 2 ? "the read stage asked for a base-quality bit of a record staged without its quality row (more than 10 low-quality bases, or no bases): such a record can never pass goodread -- the staging rule and the kernel disagree"
                                                 : "the read stage asked for the bases of a record staged without them (UZ_AUX_NO_SEQ): the selection that staged the table does not cover this batch's fetches"};
            }
            const unsigned long long used = *hused;
            c->prof[UZ_K_PHASE].last_units = (int64_t)*(const int32_t *)(hused + 1); // DNMs that took the HBM build
            if (!a.want_lists || used <= a.pool_cap) break;
            // list pool too small: grow to the exact demand and run again
            pool_cap = (size_t)used + 1024;
            st->pool.ensure(pool_cap);
            a.pool = st->pool.p; a.pool_cap = pool_cap;
        }
        // what the batch really needed: the next batch's assumption -- and the verdict on this one's
        const Sizes real = exact_sizes(bh_own);
        if (host_trace) t_sized = host_us();
        // (head room costs scratch, and a scratch that outgrows its budget halves the grid: an eighth on the linear sizes, none on the
        // pair table -- its capacity is a power of two already)
        auto room = [](int32_t v) { return (int32_t)std::min<long long>((long long)v + v / 8 + 8, 0x3FFFFFFF); };
        st->spec_caps.A = room(real.caps.A); st->spec_caps.T = room(real.caps.T); st->spec_caps.H = room(real.caps.H); st->spec_caps.C = room(real.caps.C);
        st->spec_caps.I = 4 * st->spec_caps.A;
        st->spec_caps.M = real.caps.M;
        st->spec_arena = real.arena; st->spec_arena2 = real.arena2;
        st->spec_sumP_per_dnm = 1.25 * (double)real.sumP / (double)n;
        st->spec_valid = true;
        static const bool spec_log = getenv("UZ_PHASE_SPEC_LOG") != nullptr; // development aid
        if (spec_log)
            fprintf(stderr, "[uz_launch_phase] n %d speculative %d caps A %d T %d H %d C %d M %d arena %d grid %d per_wg %zu pool %zu | real A %d T %d H %d C %d M %d arena %d\n", n,
                    (int)speculative, caps.A, caps.T, caps.H, caps.C, caps.M, arena_used, grid, per_wg, pool_cap, real.caps.A, real.caps.T, real.caps.H, real.caps.C, real.caps.M, real.arena);
        if (!speculative) break;
        const bool fits = real.caps.A <= caps.A && real.caps.T <= caps.T && real.caps.H <= caps.H && real.caps.C <= caps.C && real.caps.M <= caps.M;
        st->spec_runs++;
        if (fits) break;
        st->spec_misses++;
        speculative = false; // the batch outgrew the assumption: once more, on its own sizes
    }
#ifdef UZ_PHASE_TIMING
    {
        unsigned long long t[32];
        UZ_HIP(hipMemcpy(t, timing.p, sizeof(t), hipMemcpyDeviceToHost));
        const char *nm[17] = {"A.classify", "A.lists+C", "-", "B.overlap", "B.finish", "S.keys", "S.sort", "P.count", "P.scatter", "P.pairs", "D.finder",
                              "E.setup", "E.expand", "E.winners", "E.frontier", "F.join", "F.count"};
        unsigned long long tot = 0;
        for (int k = 0; k < 17; k++) tot += t[k];
        fprintf(stderr, "[phase timing]");
        for (int k = 0; k < 17; k++) fprintf(stderr, " %s %.1f%%", nm[k], 100.0 * (double)t[k] / (double)(tot ? tot : 1));
        fprintf(stderr, " | ticks/DNM %.0f\n", (double)tot / n);
    }
#endif
    if (status) memcpy(status, hres, (size_t)n * sizeof(int32_t));
    if (counts) memcpy(counts, hres + n, (size_t)4 * n * sizeof(int32_t));
    if (origin) memcpy(origin, hres + (size_t)5 * n, (size_t)n * sizeof(int32_t));
    if (evidence) memcpy(evidence, hres + (size_t)6 * n, (size_t)n * sizeof(int32_t));
    st->have_lists = a.want_lists != 0;
    c->phase_valid = true;
    if (host_trace)
        fprintf(stderr, "[uz] uz_launch_phase, host us: everything queued %.0f | waited for the device %.0f | exact sizes %.0f | results copied out %.0f\n", t_queued, t_synced - t_queued,
                t_sized - t_synced, host_us() - t_sized);
}

// second half of uz_phase_begin / uz_phase_end: waits for the run uz_launch_phase(defer) left in flight (or finds the results of a run
// that had to be synchronous) and hands the results out.  false: the batch outgrew the sizes it was run on (or its list pool) and
// must be run again, whole -- the caller does that with PhaseState::force_exact set.
bool uz_finish_phase(uz_ctx *c, int32_t *status, int32_t *counts, int32_t *origin, int32_t *evidence) {
    PhaseState *st = (PhaseState *)c->phase_state;
    UZ_REQUIRE(st != nullptr, UZ_E_STATE, "uz_phase_end without uz_phase_begin");
    const int32_t n = st->n;
    if (n <= 0) { c->phase_valid = true; return true; }
    int32_t *const hres = st->res_h;
    if (st->pending) {
        st->pending = false;
        UZ_HIP(hipStreamSynchronize(c->stream));
        phase_check_upload_flags(c);
        if (c->hflags[1]) {
            c->hflags[1] = 0;
            st->force_exact = true; // (the whole run states the error in full)
            return false;
        }
        unsigned long long *const hused = (unsigned long long *)(hres + (size_t)7 * n + ((7 * (size_t)n) & 1));
        const unsigned long long used = *hused;
        c->prof[UZ_K_PHASE].last_units = (int64_t)*(const int32_t *)(hused + 1);
        const Sizes real = phase_exact_sizes(reinterpret_cast<const BoundsRed *>(st->bounds_h), n);
        auto room = [](int32_t v) { return (int32_t)std::min<long long>((long long)v + v / 8 + 8, 0x3FFFFFFF); };
        st->spec_caps.A = room(real.caps.A); st->spec_caps.T = room(real.caps.T); st->spec_caps.H = room(real.caps.H); st->spec_caps.C = room(real.caps.C);
        st->spec_caps.I = 4 * st->spec_caps.A;
        st->spec_caps.M = real.caps.M;
        st->spec_arena = real.arena; st->spec_arena2 = real.arena2;
        st->spec_sumP_per_dnm = 1.25 * (double)real.sumP / (double)n;
        const Caps &caps = st->pend_caps;
        const bool fits = real.caps.A <= caps.A && real.caps.T <= caps.T && real.caps.H <= caps.H && real.caps.C <= caps.C && real.caps.M <= caps.M;
        st->spec_runs++;
        if (!fits) st->spec_misses++;
        if (!fits || (st->pend_want_lists && used > st->pend_pool_cap)) { st->force_exact = true; return false; }
        st->have_lists = st->pend_want_lists != 0;
    }
    if (status) memcpy(status, hres, (size_t)n * sizeof(int32_t));
    if (counts) memcpy(counts, hres + n, (size_t)4 * n * sizeof(int32_t));
    if (origin) memcpy(origin, hres + (size_t)5 * n, (size_t)n * sizeof(int32_t));
    if (evidence) memcpy(evidence, hres + (size_t)6 * n, (size_t)n * sizeof(int32_t));
    c->phase_valid = true;
    return true;
}

static void fetch_list_index(uz_ctx *c, PhaseState *st) {
    const size_t n = (size_t)st->n;
    st->list_start_h.resize(n);
    st->list_len_h.resize(6 * n);
    if (!n) return;
    UZ_HIP(hipMemcpyAsync(st->list_start_h.data(), st->list_start.p, n * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    UZ_HIP(hipMemcpyAsync(st->list_len_h.data(), st->list_len.p, 6 * n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    UZ_HIP(hipStreamSynchronize(c->stream));
}

// gathers lists [k0, k1) of every DNM (k in the 6-list layout) into CSR form
static int gather_lists(uz_ctx *c, int k0, int k1, int64_t *off, int32_t *val) {
    PhaseState *st = (PhaseState *)c->phase_state;
    UZ_REQUIRE(st && st->have_lists, UZ_E_STATE, "no lists kept by the last uz_phase");
    fetch_list_index(c, st);
    const int nk = k1 - k0;
    int64_t total = 0;
    for (int32_t d = 0; d < st->n; d++)
        for (int k = 0; k < nk; k++) {
            off[(size_t)nk * d + k] = total;
            if (st->list_start_h[d] >= 0) total += st->list_len_h[(size_t)6 * d + k0 + k];
        }
    off[(size_t)nk * st->n] = total;
    if (!val || total == 0) return 0;
    // copy the used part of the pool once, then slice on the host
    unsigned long long used = 0;
    UZ_HIP(hipMemcpy(&used, st->pool_cursor, sizeof(used), hipMemcpyDeviceToHost));
    std::vector<int32_t> pool((size_t)used);
    if (used) UZ_HIP(hipMemcpy(pool.data(), st->pool.p, (size_t)used * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int32_t d = 0; d < st->n; d++) {
        if (st->list_start_h[d] < 0) continue;
        long long src = st->list_start_h[d];
        for (int k = 0; k < 6; k++) {
            const int len = st->list_len_h[(size_t)6 * d + k];
            if (k >= k0 && k < k1 && len) {
                int32_t *out = val + off[(size_t)nk * d + (k - k0)];
                memcpy(out, pool.data() + src, (size_t)len * sizeof(int32_t));
                // cohort batches: query-name ids (lists 0, 1, 4, 5) are handed back relative to the kid's own table
                if (k != 2 && k != 3 && !c->phase_qbase.empty())
                    for (int x = 0; x < len; x++) out[x] -= (int32_t)c->phase_qbase[(size_t)d];
            }
            src += len;
        }
    }
    return 0;
}

int uz_phase_votes_impl(uz_ctx *c, int64_t *vote_off, int32_t *vote_val) { return gather_lists(c, 0, 4, vote_off, vote_val); }
int uz_phase_groups_impl(uz_ctx *c, int64_t *grp_off, int32_t *grp_q) { return gather_lists(c, 4, 6, grp_off, grp_q); }
