// wg.hpp -- workgroup-level building blocks of the per-DNM kernels.
//
// The per-DNM read stage is written as a sequence of block-parallel phases
// (WG_FOR loops separated by barriers) over arrays in a per-workgroup scratch
// region.  The same source compiles two ways:
//   * hipcc, gfx950: WG_FOR strides the loop over the lanes of the workgroup (a power of two),
//     WG_SYNC is __syncthreads(), scans/sorts go through LDS;
//   * -DUZ_EMU (g++, tests/emu only): one lane, loops run sequentially.  This is
//     a debugging aid for the authoring container, which has no GPU; it is never
//     loaded by the product.
// For the two builds to agree, no phase may depend on the order in which lanes
// reach an atomic: order-sensitive steps use explicit ranks, scans and sorts.
#pragma once
#include <cstdint>

#ifdef UZ_EMU
#include <algorithm>
#define UZ_DEV static inline
#define UZ_HD static inline
#define WG_NT 1
#define WG_TID 0
static inline int wg_lane_opaque() { return 0; }
#define WG_FOR(i, n) for (int i = 0; i < (int)(n); ++i)
#define WG_SYNC() ((void)0)
#define WG_T0 if (true)
UZ_DEV uint32_t wg_atomic_or(uint32_t *p, uint32_t v) { uint32_t o = *p; *p |= v; return o; }
UZ_DEV int wg_atomic_add(int *p, int v) { int o = *p; *p += v; return o; }
UZ_DEV void wg_atomic_min64(unsigned long long *p, unsigned long long v) { if (v < *p) *p = v; }
UZ_DEV unsigned long long wg_atomic_add64(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
UZ_DEV void wg_atomic_min32(uint32_t *p, uint32_t v) { if (v < *p) *p = v; }
#else
#include <hip/hip_runtime.h>
#define UZ_DEV __device__ __forceinline__
#define UZ_HD __host__ __device__ inline
// ONE WAVE PER DNM (round 5).  Rounds 1-4 gave a DNM a 256-lane workgroup: ~150 barrier-separated phases per DNM, four waves each paying a
// phase's scalar set-up, three of them idle whenever a list had fewer than 65 items (most lists of a DNM do).  With a single wavefront per
// DNM every __syncthreads() is a compiler fence (no s_barrier: the backend drops it for a workgroup that is one wave), scans are ballots,
// and a CU holds as many DNMs as its LDS has arenas for.
#define UZ_WG_NT 64
#define WG_NT UZ_WG_NT
#define WG_TID ((int)threadIdx.x)
// The lane index is re-read through an opaque move at every loop: otherwise the compiler hoists the
// per-lane address of every array (`base + 4 * tid`, a 64-bit VGPR pair each) out of ALL the phases that
// use it and keeps dozens of them alive across the whole kernel (150 VGPRs instead of < 128).
__device__ __forceinline__ int wg_lane_opaque() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
#define WG_FOR(i, n) for (int i = wg_lane_opaque(); i < (int)(n); i += WG_NT)
#define WG_SYNC() __syncthreads()
#define WG_T0 if (wg_lane_opaque() == 0) // (opaque: the mask `lane 0` is made where it is used -- hoisted, it is one more pair of scalar registers kept for the whole kernel)
UZ_DEV uint32_t wg_atomic_or(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
UZ_DEV int wg_atomic_add(int *p, int v) { return atomicAdd(p, v); }
UZ_DEV void wg_atomic_min64(unsigned long long *p, unsigned long long v) { atomicMin(p, v); }
UZ_DEV unsigned long long wg_atomic_add64(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
UZ_DEV void wg_atomic_min32(uint32_t *p, uint32_t v) { atomicMin(p, v); }
#endif
UZ_DEV void wg_atomic_min(unsigned long long *p, unsigned long long v) { wg_atomic_min64(p, v); }
UZ_DEV void wg_atomic_min(uint32_t *p, uint32_t v) { wg_atomic_min32(p, v); }
// OR into an element of a flag array: 32-bit elements directly, 8-bit elements through the aligned word that holds them (arrays start
// 16-byte aligned and are padded to 16 bytes, so the word is inside the array)
UZ_DEV void wg_or_flag(uint32_t *a, int i, uint32_t v) { wg_atomic_or(&a[i], v); }
UZ_DEV void wg_or_flag(uint8_t *a, int i, uint32_t v) {
#ifdef UZ_EMU
    a[i] = (uint8_t)(a[i] | v);
#else
    wg_atomic_or(reinterpret_cast<uint32_t *>(a + (i & ~3)), v << (8 * (i & 3)));
#endif
}

// ---- rounds of the wave: every lane takes part in every round (uniform trip count: ballots and lane scans inside the body are
// legal), `act` says whether the lane holds an item.  WG_FOR stays for loops whose bodies are lane-private.
#define WG_ROUNDS(i, n, act)                                                                                  \
    for (int i##_r0 = 0, i = wg_lane_opaque(); i##_r0 < (int)(n); i##_r0 += WG_NT, i += WG_NT)              \
        if (const bool act = i < (int)(n); true)
// rank of the lane's item among the items of the pass that satisfy `pred`, in item order: base + the number of lower lanes whose pred is
// set; base then moves on by the round's total (uniform).  Replaces flag array + scan + compaction pass.
UZ_DEV int wg_rank(bool pred, int &base) {
#ifdef UZ_EMU
    const int r = base;
    base += pred ? 1 : 0;
    return r;
#else
    const unsigned long long m = __ballot(pred);
    const int r = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    base += (int)__popcll(m);
    return r;
#endif
}
// number of lanes of the round whose pred is set (uniform)
UZ_DEV int wg_count(bool pred) {
#ifdef UZ_EMU
    return pred ? 1 : 0;
#else
    return (int)__popcll(__ballot(pred));
#endif
}
// the value of the item before the lane's own (lane - 1; lane 0: `carry`, the last item of the round before); carry then becomes the
// value of the round's last lane
UZ_DEV uint32_t wg_prev32(uint32_t v, uint32_t &carry) {
#ifdef UZ_EMU
    const uint32_t p = carry;
    carry = v;
    return p;
#else
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t up = (uint32_t)__shfl_up((int)v, 1, 64);
    const uint32_t p = lane == 0 ? carry : up;
    carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return p;
#endif
}

#ifndef UZ_EMU
// ---- cross-lane steps of a 64-lane wave without LDS traffic (gfx950) ------------------------------------------------------------
// __shfl_up / __shfl_xor compile to ds_bpermute_b32: an LDS-pipe instruction, an address computation and -- for the scans -- a lane
// predicate (`lane >= off`) per step, which the compiler computes once at the kernel's entry for every step of every inlined scan and
// sort and then keeps: in round 3's k_phase that was ~100 lane masks (200 scalar registers) spilled before the first DNM was read.
// DPP operands read another lane of the same 16-lane row inside the vector instruction itself (v_add_u32_dpp: one instruction per
// scan step, lanes without a source add 0); rows are joined by row_bcast:15 / :31, and lanes 16 / 32 apart by the gfx950 lane swaps.
#define WV_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
#define WV_ROW_SL(n) (0x100 | (n)) // lane i reads lane i + n of its row
#define WV_ROW_SR(n) (0x110 | (n)) // lane i reads lane i - n of its row
#define WV_ROW_RR(n) (0x120 | (n)) // ... lane (i - n) mod 16
#define WV_ROW_MIRROR 0x140        // lane i reads lane 15 - i of its row
#define WV_ROW_HALF_MIRROR 0x141   // lane i reads lane 7 - i of its half row
#define WV_ROW_BCAST15 0x142       // lane 15 of every row to the next row
#define WV_ROW_BCAST31 0x143       // lane 31 to the upper half
// lanes without a source (beyond the row, or masked out by row / bank mask) keep `old`
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
UZ_DEV uint32_t wv_dpp(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}
// inclusive prefix sum over the 64 lanes: six vector instructions
UZ_DEV uint32_t wv_incl_scan(uint32_t x) {
    x += wv_dpp<WV_ROW_SR(1)>(0u, x);
    x += wv_dpp<WV_ROW_SR(2)>(0u, x);
    x += wv_dpp<WV_ROW_SR(4)>(0u, x);
    x += wv_dpp<WV_ROW_SR(8)>(0u, x);
    x += wv_dpp<WV_ROW_BCAST15, 0xa>(0u, x); // rows 1 and 3 take the total of rows 0 and 2
    x += wv_dpp<WV_ROW_BCAST31, 0xc>(0u, x); // the upper half takes the total of the lower
    return x;
}
UZ_DEV int wv_incl_scan(int x) { return (int)wv_incl_scan((uint32_t)x); }
// the value of lane (i ^ J), J a power of two below 64
template <int J>
UZ_DEV uint32_t wv_xor(uint32_t x, int lane) {
    if constexpr (J == 1) return wv_dpp<WV_QUAD(1, 0, 3, 2)>(x, x);
    else if constexpr (J == 2) return wv_dpp<WV_QUAD(2, 3, 0, 1)>(x, x);
    else if constexpr (J == 4) return wv_dpp<WV_ROW_SR(4), 0xf, 0xa>(wv_dpp<WV_ROW_SL(4), 0xf, 0x5>(x, x), x); // banks 0, 2 from the right, 1, 3 from the left
    else if constexpr (J == 8) return wv_dpp<WV_ROW_RR(8)>(x, x);
    else if constexpr (J == 16) { // v_permlane16_swap: the odd rows of the first operand change places with the even rows of the second
        const auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        return (lane & 16) ? p[0] : p[1];
    } else {                      // v_permlane32_swap: the upper half of the first operand with the lower half of the second
        static_assert(J == 32, "a power of two below 64");
        const auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        return (lane & 32) ? p[0] : p[1];
    }
}
template <int J>
UZ_DEV unsigned long long wv_xor(unsigned long long x, int lane) {
    return ((unsigned long long)wv_xor<J>((uint32_t)(x >> 32), lane) << 32) | (unsigned long long)wv_xor<J>((uint32_t)x, lane);
}
// smaller and larger of a lane's value and that of lane (i ^ J): for 32-bit values and J = 16 / 32 both fall out of the swap itself
template <int J>
UZ_DEV void wv_minmax_xor(uint32_t x, int lane, uint32_t &mn, uint32_t &mx) {
    if constexpr (J == 16) {
        const auto p = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        mn = p[0] < p[1] ? p[0] : p[1]; mx = p[0] < p[1] ? p[1] : p[0];
    } else if constexpr (J == 32) {
        const auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        mn = p[0] < p[1] ? p[0] : p[1]; mx = p[0] < p[1] ? p[1] : p[0];
    } else {
        const uint32_t y = wv_xor<J>(x, lane);
        mn = x < y ? x : y; mx = x < y ? y : x;
    }
}
template <int J>
UZ_DEV void wv_minmax_xor(unsigned long long x, int lane, unsigned long long &mn, unsigned long long &mx) {
    const unsigned long long y = wv_xor<J>(x, lane);
    mn = x < y ? x : y; mx = x < y ? y : x;
}
// minimum and maximum over the 64 lanes, in every lane (uniform: the last step reads the four row results with v_readlane)
UZ_DEV void wv_allminmax(int &mn, int &mx) {
    uint32_t a = (uint32_t)mn, b = (uint32_t)mx;
#define WV_STEP(CTRL)                                                           \
    {                                                                           \
        const int a2 = (int)wv_dpp<CTRL>(a, a), b2 = (int)wv_dpp<CTRL>(b, b);   \
        a = (uint32_t)(a2 < (int)a ? a2 : (int)a);                              \
        b = (uint32_t)(b2 > (int)b ? b2 : (int)b);                              \
    }
    WV_STEP(WV_QUAD(1, 0, 3, 2)) WV_STEP(WV_QUAD(2, 3, 0, 1)) WV_STEP(WV_ROW_HALF_MIRROR) WV_STEP(WV_ROW_MIRROR)
#undef WV_STEP
    int lo = __builtin_amdgcn_readlane((int)a, 0), hi = __builtin_amdgcn_readlane((int)b, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const int a2 = __builtin_amdgcn_readlane((int)a, r), b2 = __builtin_amdgcn_readlane((int)b, r);
        lo = a2 < lo ? a2 : lo;
        hi = b2 > hi ? b2 : hi;
    }
    mn = lo; mx = hi;
}
#endif

#ifndef WG_SORT_LDS_CAP
#define WG_SORT_LDS_CAP 256 // u64 keys copied into the 2 KiB LDS sort buffer; larger sorts run in place (LDS arena or HBM scratch)
#endif

// SORT_CAP: room for the keys of a small sort whose array is NOT in LDS (wg_sort64); the kernel build whose arrays all are
// in LDS sorts in place and keeps the 2 KiB for its arena
template <int SORT_CAP>
struct WgSharedT {
    int part[WG_NT + 1];
    unsigned long long sortbuf[SORT_CAP];
    int bcast[4];
#ifdef UZ_PHASE_TIMING
    unsigned long long tick[24]; // diagnostic build: shader-clock ticks per phase of this wave's DNMs
#endif
};
typedef WgSharedT<WG_SORT_LDS_CAP> WgShared;

// In-place exclusive scan of a[0..n) -> returns the total.  Block-uniform call.
// Each lane sums a contiguous chunk, lanes are combined with wave shuffles (no barrier) and the
// waves through WG_NT/64 words of LDS: three barriers per call.
template <typename SH>
UZ_DEV int wg_exscan(int *a, int n, SH *sh) {
#ifdef UZ_EMU
    int s = 0;
    for (int i = 0; i < n; i++) { int v = a[i]; a[i] = s; s += v; }
    return s;
#else
    __syncthreads();
    const int t = wg_lane_opaque();
    const int chunk = (n + WG_NT - 1) / WG_NT;
    int lo = t * chunk; if (lo > n) lo = n;
    int hi = lo + chunk; if (hi > n) hi = n;
    int s = 0;
    for (int i = lo; i < hi; i++) s += a[i];
    const int incl = wv_incl_scan(s);
    if ((t & 63) == 63) sh->part[t >> 6] = incl;
    __syncthreads();
    int wave_prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WG_NT / 64; w++) {
        const int v = sh->part[w];
        if (w < (t >> 6)) wave_prefix += v;
        total += v;
    }
    int run = wave_prefix + incl - s;
    for (int i = lo; i < hi; i++) { const int v = a[i]; a[i] = run; run += v; }
    __syncthreads();
    return total;
#endif
}

// Lane-chunk helpers: lane t of the workgroup owns the contiguous slice [lo, hi) of 0..n.
UZ_DEV void wg_chunk(int n, int &lo, int &hi) {
#ifdef UZ_EMU
    lo = 0; hi = n;
#else
    const int chunk = (n + WG_NT - 1) / WG_NT;
    lo = (int)threadIdx.x * chunk; if (lo > n) lo = n;
    hi = lo + chunk; if (hi > n) hi = n;
#endif
}

// Exclusive scan ACROSS LANES of K per-lane values at once (one pair of barriers for all K):
// off[k] = sum of c[k] over lower lanes, tot[k] = sum over all lanes.  Block-uniform call.
template <int K, typename SH>
UZ_DEV void wg_lane_exscan(const int (&c)[K], int (&off)[K], int (&tot)[K], SH *sh) {
#ifdef UZ_EMU
    for (int k = 0; k < K; k++) { off[k] = 0; tot[k] = c[k]; }
#else
    static_assert(K * (WG_NT / 64) <= WG_NT + 1, "part[] too small");
    const int t = wg_lane_opaque(), lane = t & 63, wv = t >> 6;
    int incl[K];
#pragma unroll
    for (int k = 0; k < K; k++) incl[k] = wv_incl_scan(c[k]);
    __syncthreads(); // part[] may still be read by a previous scan
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < K; k++) sh->part[k * (WG_NT / 64) + wv] = incl[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        int pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < WG_NT / 64; w++) {
            const int v = sh->part[k * (WG_NT / 64) + w];
            if (w < wv) pre += v;
            all += v;
        }
        off[k] = pre + incl[k] - c[k];
        tot[k] = all;
    }
#endif
}

// Block-wide minimum and maximum of per-lane partial results (two barriers, no atomics).
template <typename SH>
UZ_DEV void wg_minmax(int lmin, int lmax, int &mn, int &mx, SH *sh) {
#ifdef UZ_EMU
    mn = lmin; mx = lmax;
#else
    const int t = wg_lane_opaque(), lane = t & 63, wv = t >> 6;
    wv_allminmax(lmin, lmax);
    __syncthreads();
    if (lane == 0) { sh->part[wv] = lmin; sh->part[WG_NT / 64 + wv] = lmax; }
    __syncthreads();
    mn = sh->part[0]; mx = sh->part[WG_NT / 64];
#pragma unroll
    for (int w = 1; w < WG_NT / 64; w++) {
        const int a = sh->part[w], b = sh->part[WG_NT / 64 + w];
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    __syncthreads();
#endif
}

// Ascending sort of a[0..n) (distinct keys).  a must have room for the next power of two.
// a_in_lds: the caller knows a[] lies in the workgroup's LDS arena; the stages then run on an
// LDS-typed pointer (ds_read/ds_write) instead of flat accesses.
#ifndef UZ_EMU
template <typename P>
UZ_DEV void wg_bitonic_stages(P w, int N) {
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < N; i += WG_NT) {
                const int x = i ^ j;
                if (x > i) {
                    const auto u = w[i], v = w[x];
                    const bool up = (i & k) == 0;
                    if ((u > v) == up) { w[i] = v; w[x] = u; }
                }
            }
            __syncthreads();
        }
    }
}
#endif
#ifndef UZ_EMU
// Bitonic sort with the keys held in registers: lane t owns elements t + WG_NT * r.  Partners at
// distance >= WG_NT are other registers of the same lane; partners at distance < 64 are other lanes of the wave, reached by DPP
// operands (1, 2, 4, 8) and the gfx950 lane swaps (16, 32) -- no LDS traffic; only the distances in between (64 and 128 for a 256-lane
// workgroup) go through LDS and a barrier.  buf: LDS, N entries.  N = R * WG_NT.
// The loop over the merge sizes k is NOT unrolled and re-reads the lane index through an opaque move: unrolled, every one of the 45
// stages of a 1024-key sort owns a pair of lane masks ("lower half of the exchange", "ascending block") that the compiler computes at the
// kernel's entry and keeps -- for three inlined sorts that was ~100 masks in 200 spilled scalar registers, fetched back with
// v_readlane in front of every exchange.  Rolled, a merge's masks are made where they are used (two instructions) and the body is a
// sixth of the code.
template <int J, int R, typename T>
UZ_DEV void wg_bitonic_wave_stage(T (&v)[R], const int (&kdesc)[R], int t) {
    const bool lower = (t & J) == 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        T mn, mx;
        wv_minmax_xor<J>(v[r], t, mn, mx);
        v[r] = (lower == (kdesc[r] == 0)) ? mn : mx;
    }
}
template <int R, typename T = unsigned long long>
UZ_DEV void wg_bitonic_regs(__attribute__((address_space(3))) T *buf, T *a, int n) {
    int t = threadIdx.x;
    T v[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const int i = t + WG_NT * r; v[r] = i < n ? a[i] : (T) ~(T)0; }
    constexpr int N = R * WG_NT;
#pragma nounroll
    for (int k = 2; k <= N; k <<= 1) {
        asm volatile("" : "+v"(t)); // (see above)
        int kdesc[R]; // element r lies in a descending block of this merge
#pragma unroll
        for (int r = 0; r < R; r++) kdesc[r] = (t + WG_NT * r) & k;
#pragma unroll
        for (int rr = R / 2; rr >= 1; rr >>= 1) {
            if (rr * WG_NT < k) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (!(r & rr)) {
                        const bool up = kdesc[r] == 0;
                        const T x = v[r], y = v[r | rr];
                        if ((x > y) == up) { v[r] = y; v[r | rr] = x; }
                    }
                }
            }
        }
        for (int j = (k >> 1) < (WG_NT >> 1) ? (k >> 1) : (WG_NT >> 1); j >= 64; j >>= 1) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; r++) buf[t + WG_NT * r] = v[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; r++) {
                const T y = buf[(t ^ j) + WG_NT * r];
                const bool up = kdesc[r] == 0, lower = (t & j) == 0;
                const T x = v[r];
                v[r] = (lower == up) ? (x < y ? x : y) : (x > y ? x : y);
            }
        }
        if (k > 32) wg_bitonic_wave_stage<32, R, T>(v, kdesc, t);
        if (k > 16) wg_bitonic_wave_stage<16, R, T>(v, kdesc, t);
        if (k > 8) wg_bitonic_wave_stage<8, R, T>(v, kdesc, t);
        if (k > 4) wg_bitonic_wave_stage<4, R, T>(v, kdesc, t);
        if (k > 2) wg_bitonic_wave_stage<2, R, T>(v, kdesc, t);
        wg_bitonic_wave_stage<1, R, T>(v, kdesc, t);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; r++) { const int i = t + WG_NT * r; if (i < n) a[i] = v[r]; }
    __syncthreads();
}
#endif
template <typename SH>
UZ_DEV void wg_sort64(unsigned long long *a, int n, SH *sh, bool a_in_lds = false) {
#ifdef UZ_EMU
    (void)sh; (void)a_in_lds;
    std::sort(a, a + n);
#else
    typedef __attribute__((address_space(3))) unsigned long long *lds_u64;
    __syncthreads();
    if (n <= 1) return;
    int N = 1;
    while (N < n) N <<= 1;
    constexpr int sort_cap = (int)(sizeof(sh->sortbuf) / sizeof(unsigned long long));
    const bool in_buf = !a_in_lds && N <= sort_cap;
    if (in_buf || (a_in_lds && N <= 4 * WG_NT)) { // 8 keys per lane would push the kernel past 128 VGPRs
        lds_u64 buf = in_buf ? (lds_u64)sh->sortbuf : (lds_u64)a;
        if (N <= WG_NT) wg_bitonic_regs<1>(buf, a, n);
        else if (N <= 2 * WG_NT) wg_bitonic_regs<2>(buf, a, n);
        else wg_bitonic_regs<4>(buf, a, n);
        return;
    }
    for (int i = n + (int)threadIdx.x; i < N; i += WG_NT) a[i] = ~0ULL;
    __syncthreads();
    if (a_in_lds) wg_bitonic_stages((lds_u64)a, N);
    else wg_bitonic_stages(a, N);
#endif
}

// The same for 32-bit keys that lie in the workgroup's LDS arena (the arena build of the per-DNM kernel packs its pair-table keys into
// 32 bits: half the shuffles and one min / max per exchange instead of a 64-bit compare and two selects).
UZ_DEV void wg_sort32_lds(uint32_t *a, int n) {
#ifdef UZ_EMU
    std::sort(a, a + n);
#else
    typedef __attribute__((address_space(3))) uint32_t *lds_u32;
    __syncthreads();
    if (n <= 1) return;
    int N = 1;
    while (N < n) N <<= 1;
    if (N <= 4 * WG_NT) {
        if (N <= WG_NT) wg_bitonic_regs<1, uint32_t>((lds_u32)a, a, n);
        else if (N <= 2 * WG_NT) wg_bitonic_regs<2, uint32_t>((lds_u32)a, a, n);
        else wg_bitonic_regs<4, uint32_t>((lds_u32)a, a, n);
        return;
    }
#if WG_NT == 64
    // one wave per DNM: a lane holds up to 16 keys, so 1024 keys are sorted without touching LDS between the first load and the last store
    if (N <= 16 * WG_NT) {
        if (N <= 8 * WG_NT) wg_bitonic_regs<8, uint32_t>((lds_u32)a, a, n);
        else wg_bitonic_regs<16, uint32_t>((lds_u32)a, a, n);
        return;
    }
#endif
    for (int i = n + (int)threadIdx.x; i < N; i += WG_NT) a[i] = ~0u;
    __syncthreads();
    wg_bitonic_stages((lds_u32)a, N);
#endif
}
