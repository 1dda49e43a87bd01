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
#else
#include <hip/hip_runtime.h>
#define UZ_DEV __device__ __forceinline__
#define UZ_HD __host__ __device__ inline
#ifndef UZ_WG_NT
#define UZ_WG_NT 256 // lanes per DNM workgroup: 256 lanes, 28 KiB LDS arena, 5 workgroups per CU measured best on MI355X
#endif
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
#define WG_T0 if (threadIdx.x == 0)
UZ_DEV uint32_t wg_atomic_or(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
UZ_DEV int wg_atomic_add(int *p, int v) { return atomicAdd(p, v); }
UZ_DEV void wg_atomic_min64(unsigned long long *p, unsigned long long v) { atomicMin(p, v); }
UZ_DEV unsigned long long wg_atomic_add64(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
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
    const int t = threadIdx.x;
    const int chunk = (n + WG_NT - 1) / WG_NT;
    int lo = t * chunk; if (lo > n) lo = n;
    int hi = lo + chunk; if (hi > n) hi = n;
    int s = 0;
    for (int i = lo; i < hi; i++) s += a[i];
    int incl = s;
    const int lane = t & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) sh->part[t >> 6] = incl;
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
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int incl[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        int v = c[k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        incl[k] = v;
    }
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
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int a = __shfl_xor(lmin, o, 64), b = __shfl_xor(lmax, o, 64);
        lmin = a < lmin ? a : lmin;
        lmax = b > lmax ? b : lmax;
    }
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
// distance >= WG_NT are other registers of the same lane, partners at distance < 64 are reached by a
// wave shuffle; only the distances in between (64 and 128 for a 256-lane workgroup) go through LDS and a
// barrier.  buf: LDS, N entries.  N = R * WG_NT.
template <int R, typename T = unsigned long long>
UZ_DEV void wg_bitonic_regs(__attribute__((address_space(3))) T *buf, T *a, int n) {
    const int t = threadIdx.x;
    T v[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const int i = t + WG_NT * r; v[r] = i < n ? a[i] : (T) ~(T)0; }
    const int N = R * WG_NT;
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int rr = R / 2; rr >= 1; rr >>= 1) {
            if (rr * WG_NT < k) {
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (!(r & rr)) {
                        const bool up = ((t + WG_NT * r) & k) == 0;
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
                const bool up = ((t + WG_NT * r) & k) == 0, lower = (t & j) == 0;
                const T x = v[r];
                v[r] = (lower == up) ? (x < y ? x : y) : (x > y ? x : y);
            }
        }
        for (int j = (k >> 1) < 32 ? (k >> 1) : 32; j >= 1; j >>= 1) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const T x = v[r];
                const T y = __shfl_xor(x, j, 64);
                const bool up = ((t + WG_NT * r) & k) == 0, lower = (t & j) == 0;
                v[r] = (lower == up) ? (x < y ? x : y) : (x > y ? x : y);
            }
        }
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
    for (int i = n + (int)threadIdx.x; i < N; i += WG_NT) a[i] = ~0u;
    __syncthreads();
    wg_bitonic_stages((lds_u32)a, N);
#endif
}
