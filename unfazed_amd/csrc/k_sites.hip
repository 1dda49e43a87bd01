// k_sites.hip -- site stage on gfx950.
//
//   K1  k_site_scan     one streaming pass over a trio's genotype columns -> one class
//                       byte per site (UZ_CL_*).  HBM-bound: 19 B read + 1 B written per site
//                       (gt 1 + 9 x u16 columns; DESIGN.md "K1").  Each lane owns SPT consecutive
//                       sites so every column is read with 16-byte (SPT=8) or 2x16-byte (SPT=16)
//                       fully coalesced loads, all issued before the first use.
//   K2  k_window        per-DNM window bounds (two binary searches on pos[]) and ordered
//                       emission of the candidate / het lists (count pass, scan, fill pass).
//
// Logic restated from reference unfazed/informative_site_finder.py: is_high_quality_site
// (:46-73), get_kid_allele (:76-134), the per-variant body of find (:239-339, duplicated at
// :442-543) and get_position (:10-43).  No MFMA: there is no contraction here.
#include "uz_ctx.hpp"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

struct SiteParams {
    int32_t min_gt_qual, min_depth;
    double lo_ref, hi_ref, lo_het, hi_het, lo_alt, hi_alt; // allele-balance windows per genotype
};

// The 16-bit columns hold 0..32767, or 0xFFFF for cyvcf2's -1 (missing): read as SIGNED halfwords the
// sentinel decodes itself (one v_bfe_i32 / v_ashrrev per value instead of and + compare + select).
__device__ __forceinline__ int dec16(uint32_t v) { return (int)(int16_t)(uint16_t)v; }

// Allele-balance window test without a division on the hot path.  For a fixed total depth t the
// correctly rounded quotient RN(a / t) is monotone in the integer a, so the set of alt depths a with
//     lo <= RN(a / t) <= hi
// is an interval [amin[t], amax[t]].  k_build_ab_lut finds it once per parameter set with the SAME
// IEEE f64 division the reference's numpy expression performs (informative_site_finder.py:69-71),
// for every total the columns can produce: t = rd + ad in [-2, 65534].  The table has one row per
// genotype code (row 2 = unknown genotype: always empty, :62-63) and already folds in the depth test
// (t < min_depth -> empty, :66), so is_high_quality_site is  gq >= min  &  amin <= ad <= amax.
// t == 0 (0/0 -> NaN, +-1/0 -> +-inf) is an interval unless BOTH infinities pass (only with infinite
// thresholds); that case is flagged and served by the kernel variant with an explicit t == 0 test.
// A 30x genome touches a few hundred bytes of the 2 MiB table (L1-resident).
#define UZ_AB_T_MIN (-2)
#define UZ_AB_T_MAX 65534
#define UZ_AB_LUT_N (UZ_AB_T_MAX - UZ_AB_T_MIN + 1)
#define UZ_AB_A_MIN (-1)
#define UZ_AB_A_MAX 32767
__device__ __forceinline__ double ab_lo(const SiteParams &P, int gt) { return gt == UZ_HOM_REF ? P.lo_ref : (gt == UZ_HOM_ALT ? P.lo_alt : P.lo_het); }
__device__ __forceinline__ double ab_hi(const SiteParams &P, int gt) { return gt == UZ_HOM_REF ? P.hi_ref : (gt == UZ_HOM_ALT ? P.hi_alt : P.hi_het); }

#define UZ_AB_LDS_T 512 // totals below this (minus UZ_AB_T_MIN) are served from the LDS copy of the table
__global__ void k_build_ab_lut(SiteParams P, int2 *lut, uint32_t *small, int *t0_special) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * UZ_AB_LUT_N) return;
    const int gt = i / UZ_AB_LUT_N, t = i % UZ_AB_LUT_N + UZ_AB_T_MIN;
    const double lo = ab_lo(P, gt), hi = ab_hi(P, gt);
    int2 r = make_int2(0x7FFFFFFF, -0x7FFFFFFF); // empty interval
    if (gt != UZ_GT_UNKNOWN && t >= P.min_depth) {
        if (t > 0) {
            // a/t is increasing in a
            int l = UZ_AB_A_MIN, h = UZ_AB_A_MAX + 1; // smallest a with lo <= RN(a/t); A_MAX + 1 = none
            while (l < h) { const int m = l + ((h - l) >> 1); if (lo <= (double)m / (double)t) h = m; else l = m + 1; }
            const int amin = l;
            l = UZ_AB_A_MIN - 1; h = UZ_AB_A_MAX; // largest a with RN(a/t) <= hi; A_MIN - 1 = none
            while (l < h) { const int m = l + ((h - l + 1) >> 1); if ((double)m / (double)t <= hi) l = m; else h = m - 1; }
            if (amin <= UZ_AB_A_MAX && l >= UZ_AB_A_MIN) r = make_int2(amin, l);
        } else if (t < 0) {
            // only a in {-1, 0} can give a negative total (one or both depths missing)
            int amin = 0x7FFFFFFF, amax = -0x7FFFFFFF;
            for (int a = -1; a <= 0; a++) {
                const double ab = (double)a / (double)t;
                if (lo <= ab && ab <= hi) { if (a < amin) amin = a; if (a > amax) amax = a; }
            }
            r = make_int2(amin, amax); // any subset of {-1, 0} is an interval
        } else {
            // t == 0: a = 0 -> NaN (never passes), a = +-1 -> +-inf
            const double pinf = 1.0 / 0.0, ninf = -1.0 / 0.0;
            const bool neg = lo <= ninf && ninf <= hi, pos = lo <= pinf && pinf <= hi;
            if (neg && pos) { r = make_int2(-1, 1); atomicOr(t0_special, 1); } // not an interval: a = 0 must fail
            else if (neg) r = make_int2(-1, -1);
            else if (pos) r = make_int2(1, 1);
        }
    }
    lut[i] = r;
    if (t - UZ_AB_T_MIN < UZ_AB_LDS_T) { // packed copy for LDS: amin | amax << 16 as signed halfwords
        const int lo16 = r.x > 32767 ? 32767 : r.x, hi16 = r.y < -32768 ? -32768 : r.y;
        small[gt * UZ_AB_LDS_T + (t - UZ_AB_T_MIN)] = ((uint32_t)lo16 & 0xFFFFu) | ((uint32_t)hi16 << 16);
    }
}

// is_high_quality_site (:46-73)
template <bool T0>
__device__ __forceinline__ bool hq(const SiteParams &P, const int2 *__restrict__ lut, const uint32_t *lds, int gt, int rd, int ad, int gq) {
    const int t = rd + ad;
    const uint32_t ti = (uint32_t)(t - UZ_AB_T_MIN);
    int amin, amax;
    if (ti < UZ_AB_LDS_T) { // the common case: a conflict-light LDS read instead of a 64-address gather through L1
        const uint32_t e = lds[gt * UZ_AB_LDS_T + ti];
        amin = (int)(int16_t)(e & 0xFFFFu);
        amax = (int)(int16_t)(e >> 16);
    } else {
        const int2 b = lut[gt * UZ_AB_LUT_N + ti];
        amin = b.x; amax = b.y;
    }
    bool ok = (gq >= P.min_gt_qual) & (ad >= amin) & (ad <= amax);
    if (T0) ok &= !((t == 0) & (ad == 0));
    return ok;
}

// the same test on depths of any size, evaluated directly (the sites the 16-bit columns cannot hold: uz_family_view.wide_*): the
// reference's own order -- genotype, GQ, depth, then the f64 quotient against the interval (:62-71)
__device__ __forceinline__ bool hq_wide(const SiteParams &P, int gt, int rd, int ad, int gq) {
    if (gt == UZ_GT_UNKNOWN) return false;
    const long long t = (long long)rd + (long long)ad;
    if (gq < P.min_gt_qual || t < (long long)P.min_depth) return false;
    const double ab = (double)ad / (double)t; // 0 / 0 -> NaN (fails both comparisons), x / 0 -> +-inf: what numpy gives the reference
    return ab_lo(P, gt) <= ab && ab <= ab_hi(P, gt);
}

// parental pattern (:307-320) as a 16 x 2-bit table indexed by dad | mom << 2: 1 = alt_parent is dad, 2 = mom
#define UZ_PATTERN_TABLE ((1u << 2) | (1u << 6) | (2u << 8) | (2u << 24) | (1u << 14) | (2u << 26))

// CNV = false: SNV / breakpoint mode only (class bits HET, CAND, ALT_DAD) -- what find(...,
// whole_region=False) evaluates (:292-295); CNV = true adds the DEL / DUP codes of get_kid_allele,
// which only find(..., whole_region=True) reaches (:286-291).
template <bool CNV, bool T0, bool WIDE = false>
__device__ __forceinline__ uint8_t classify_site(const SiteParams &P, const int2 *__restrict__ lut, const uint32_t *lds, uint32_t g, int rdk,
                                                 int adk, int gqk, int rdd, int add, int gqd, int rdm, int adm, int gqm) {
    const int kid = g & 3, dad = (g >> 2) & 3, mom = (g >> 4) & 3;
    const bool hqk = WIDE ? hq_wide(P, kid, rdk, adk, gqk) : hq<T0>(P, lut, lds, kid, rdk, adk, gqk);
    const bool hqd = WIDE ? hq_wide(P, dad, rdd, add, gqd) : hq<T0>(P, lut, lds, dad, rdd, add, gqd);
    const bool hqm = WIDE ? hq_wide(P, mom, rdm, adm, gqm) : hq<T0>(P, lut, lds, mom, rdm, adm, gqm);
    const uint32_t pcode = (UZ_PATTERN_TABLE >> (((g >> 2) & 15u) * 2)) & 3u;
    const bool pattern = pcode != 0, alt_dad = pcode == 1;
    uint32_t c = 0;
    if (kid == UZ_HET && hqd && hqm) c |= UZ_CL_HET; // :268-284
    if (alt_dad) c |= UZ_CL_ALT_DAD;
    if (hqd && hqm && pattern) {
        if (kid == UZ_HET && hqk) c |= UZ_CL_CAND; // :292-295
        if constexpr (CNV) {
        // hemizygous unique-allele check :324-337
        bool unique = true;
        if (kid == UZ_HOM_ALT || kid == UZ_HOM_REF) {
            const bool het_in = (dad == UZ_HET) || (mom == UZ_HET);
            const bool hom_in = (dad != UZ_HET && dad != UZ_GT_UNKNOWN) || (mom != UZ_HET && mom != UZ_GT_UNKNOWN);
            if (het_in && hom_in) {
                if ((dad == UZ_HOM_ALT || dad == UZ_HOM_REF) && kid == dad) unique = false;
                if ((mom == UZ_HOM_ALT || mom == UZ_HOM_REF) && kid == mom) unique = false;
            }
        }
        if (unique) {
            // get_kid_allele :76-134 for a DEL and for a DUP
            uint32_t kdel = UZ_KA_NONE, kdup = UZ_KA_NONE;
            if ((rdk + adk) > 4) { // :80
                if (kid == UZ_HOM_ALT) kdel = UZ_KA_REF_PARENT;
                else if (kid == UZ_HOM_REF) kdel = UZ_KA_ALT_PARENT;
            }
            if (rdk > 2 && adk > 2 && (rdk + adk) > P.min_depth && kid == UZ_HET) { // :89-97 (a few % of the sites)
                const double abk = (double)adk / (double)(rdk + adk);
                const double abd = (double)add / (double)(rdd + add);
                const double abm = (double)adm / (double)(rdm + adm);
                const double s = abd + abm;
                const bool shared_dup = ((s < 1.0) && (abk > 0.5)) || ((s > 1.0) && (abk < 0.5)); // :110-116
                if (!shared_dup) {
                    if (abk >= 0.67) kdup = UZ_KA_ALT_PARENT;      // :119-121
                    else if (abk <= 0.33) kdup = UZ_KA_REF_PARENT; // :122-124
                }
            }
            c |= kdel << UZ_CL_DEL_SHIFT;
            c |= kdup << UZ_CL_DUP_SHIFT;
        }
        }
    }
    return (uint8_t)c;
}

template <int SPT>
struct ColVec; // SPT u16 values
template <>
struct ColVec<8> {
    uint4 v;
#ifdef UZ_K1_NT // (variant build: the columns are read once -- non-temporal loads)
    __device__ __forceinline__ void load(const uint16_t *p) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        v.x = __builtin_nontemporal_load(q); v.y = __builtin_nontemporal_load(q + 1); v.z = __builtin_nontemporal_load(q + 2); v.w = __builtin_nontemporal_load(q + 3);
    }
#else
    __device__ __forceinline__ void load(const uint16_t *p) { v = *reinterpret_cast<const uint4 *>(p); }
#endif
    __device__ __forceinline__ uint32_t get(int i) const {
        const uint32_t w = (&v.x)[i >> 1];
        return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
    }
};
template <>
struct ColVec<16> {
    uint4 v[2];
    __device__ __forceinline__ void load(const uint16_t *p) {
        v[0] = reinterpret_cast<const uint4 *>(p)[0];
        v[1] = reinterpret_cast<const uint4 *>(p)[1];
    }
    __device__ __forceinline__ uint32_t get(int i) const {
        const uint32_t w = (&v[i >> 3].x)[(i & 7) >> 1];
        return (i & 1) ? (w >> 16) : (w & 0xFFFFu);
    }
};

struct FamPtrs {
    const uint8_t *gt;
    const uint16_t *rd[3], *ad[3], *gq[3];
};

// A cohort scan (BATCH) classifies many trios of one sites table in a single launch: blockIdx.y picks
// the family, so one pass of 288 GB-class genotype columns costs one launch ramp instead of hundreds.
struct FamBatchItem {
    FamPtrs f;
    uint8_t *cls;
};

template <int SPT, bool CNV, bool T0, bool BATCH>
__global__ __launch_bounds__(256) void k_site_scan(FamPtrs f1, uint8_t *cls1, int64_t n, SiteParams Pk, const int2 *__restrict__ lut,
                                                   const uint32_t *__restrict__ small, const FamBatchItem *__restrict__ batch) {
    FamPtrs f = f1;
    uint8_t *__restrict__ cls = cls1;
    if constexpr (BATCH) { f = batch[blockIdx.y].f; cls = batch[blockIdx.y].cls; }
    // every workgroup keeps the low-depth part of the threshold table in LDS (8 KiB) and walks the
    // site chunks grid-stride, so the fill is paid once per workgroup, not per chunk
    __shared__ uint32_t lds_lut[4 * UZ_AB_LDS_T];
    for (int i = threadIdx.x; i < 4 * UZ_AB_LDS_T; i += 256) lds_lut[i] = small[i];
    __syncthreads();
    // Read the thresholds into registers up front.  Left in the kernarg struct, `c ? P.a : P.b` is
    // compiled as a select of ADDRESSES followed by a per-lane global load; on copies it is a v_cndmask.
    const SiteParams P = {Pk.min_gt_qual, Pk.min_depth, Pk.lo_ref, Pk.hi_ref, Pk.lo_het, Pk.hi_het, Pk.lo_alt, Pk.hi_alt};
    const int64_t n_chunks = (n + 256 * SPT - 1) / (256 * SPT);
    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int64_t base = (chunk * 256 + threadIdx.x) * SPT;
    if (base >= n) continue;
    if (base + SPT <= n) {
        // issue all 19-20 vector loads before the first use
        ColVec<SPT> c[9];
        uint32_t gw[SPT / 4];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            c[m].load(f.rd[m] + base);
            c[3 + m].load(f.ad[m] + base);
            c[6 + m].load(f.gq[m] + base);
        }
        if constexpr (SPT == 16) {
            const uint4 g = *reinterpret_cast<const uint4 *>(f.gt + base);
            gw[0] = g.x; gw[1] = g.y; gw[2] = g.z; gw[3] = g.w;
        } else {
            const uint2 g = *reinterpret_cast<const uint2 *>(f.gt + base);
            gw[0] = g.x; gw[1] = g.y;
        }
        uint32_t out[SPT / 4];
#pragma unroll
        for (int w = 0; w < SPT / 4; w++) {
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int i = w * 4 + b;
                const uint32_t g = (gw[w] >> (8 * b)) & 0xFFu;
                uint32_t cl = 0;
                if (!(g & 0x40u)) // bit 6: complex record (:239-244)
                    cl = classify_site<CNV, T0>(P, lut, lds_lut, g, dec16(c[0].get(i)), dec16(c[3].get(i)), dec16(c[6].get(i)),
                                       dec16(c[1].get(i)), dec16(c[4].get(i)), dec16(c[7].get(i)),
                                       dec16(c[2].get(i)), dec16(c[5].get(i)), dec16(c[8].get(i)));
                o |= cl << (8 * b);
            }
            out[w] = o;
        }
        if constexpr (SPT == 16)
            *reinterpret_cast<uint4 *>(cls + base) = make_uint4(out[0], out[1], out[2], out[3]);
        else
            *reinterpret_cast<uint2 *>(cls + base) = make_uint2(out[0], out[1]);
    } else {
        for (int64_t i = base; i < n; i++) {
            const uint32_t g = f.gt[i];
            uint8_t cl = 0;
            if (!(g & 0x40u))
                cl = classify_site<CNV, T0>(P, lut, lds_lut, g, dec16(f.rd[0][i]), dec16(f.ad[0][i]), dec16(f.gq[0][i]), dec16(f.rd[1][i]),
                                   dec16(f.ad[1][i]), dec16(f.gq[1][i]), dec16(f.rd[2][i]), dec16(f.ad[2][i]),
                                   dec16(f.gq[2][i]));
            cls[i] = cl;
        }
    }
    } // chunk loop
}

// the sites listed as too deep for the 16-bit columns: their class bytes again, from the 32-bit depths (one lane per site; a handful)
template <bool CNV>
__global__ void k_site_scan_wide(FamPtrs f, uint8_t *cls, int64_t n_wide, const int64_t *__restrict__ site, const int32_t *__restrict__ dep,
                                 SiteParams P) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_wide) return;
    const int64_t i = site[k];
    const uint32_t g = f.gt[i];
    uint8_t cl = 0;
    if (!(g & 0x40u))
        cl = classify_site<CNV, false, true>(P, nullptr, nullptr, g, dep[k], dep[3 * n_wide + k], dec16(f.gq[0][i]), dep[n_wide + k], dep[4 * n_wide + k],
                                             dec16(f.gq[1][i]), dep[2 * n_wide + k], dep[5 * n_wide + k], dec16(f.gq[2][i]));
    cls[i] = cl;
}

__device__ __forceinline__ int64_t lower_bound(const int32_t *a, int64_t lo, int64_t hi, int64_t v) {
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// lower bound by a whole wave: 64 probes per round (a 64-ary search: four rounds over 20 M sites instead of 24 dependent
// loads); every lane of the wave must call it with the same arguments
__device__ __forceinline__ int64_t lower_bound_wave(const int32_t *__restrict__ a, int64_t lo, int64_t hi, int64_t v, int lane) {
    while (hi - lo > 64) {
        const int64_t step = (hi - lo + 63) >> 6;
        const int64_t idx = lo + (int64_t)lane * step;
        const bool lt = idx < hi && (int64_t)a[idx] < v; // sorted: true for a prefix of the lanes
        const int k = __popcll(__ballot(lt));
        if (k == 0) return lo;
        const int64_t nhi = lo + (int64_t)k * step;
        lo = lo + (int64_t)(k - 1) * step + 1;
        hi = nhi < hi ? nhi : hi;
    }
    const int64_t idx = lo + lane;
    const bool lt = idx < hi && (int64_t)a[idx] < v;
    return lo + __popcll(__ballot(lt));
}

struct WinArgs {
    int32_t n;
    const int32_t *contig, *start, *end;
    const uint8_t *vartype, *mult;
    const int64_t *contig_off;
    int32_t n_contigs;
    const int32_t *pos;
    const uint8_t *cls;
    const uint8_t *const *cls_of; // cohort batches: class column per family ...
    const int32_t *fam_idx;       // ... and the family of every DNM (null: `cls` for all)
    int64_t sd;
    int mode;
    int64_t *range; // [4n] site index range of each of a DNM's (up to two) windows: found by the count pass, reused by the fill pass
    int64_t cap_c, cap_h; // fill pass: what the candidate / het lists hold -- a DNM whose slice ends beyond it writes nothing (uz_launch_find's first try)
};

// One lane walks the sites of one DNM in the reference's order: by position, window-1 copy before
// window-2 copy (= stable sort of the concatenated windows, :341-342), each entry `mult` times.  Only the rare DNM whose
// two windows overlap still goes this way (k_window_wave below takes every other).
template <bool FILL>
__device__ void window_serial(const WinArgs &a, int32_t d, const uint8_t *__restrict__ cls, int64_t clo, int64_t chi, int64_t w[2][2], int nw,
                              int64_t oc, int64_t oh, int32_t *cand_idx, uint8_t *cand_flags, int32_t *het_idx, int64_t &nc, int64_t &nh) {
    const int64_t st = a.start[d], en = a.end[d];
    const bool whole = a.mode & UZ_FIND_WHOLE_REGION;
    const int vt = a.vartype[d];
    const int mult = a.mult[d];
    const bool small_event = (en - st) < 20;
    int64_t i = lower_bound(a.pos, clo, chi, w[0][0] - 1);
    const int64_t hi_all = lower_bound(a.pos, clo, chi, w[nw - 1][1]);
    while (i < hi_all) {
        const int32_t p = a.pos[i];
        int64_t j = i + 1;
        while (j < hi_all && a.pos[j] == p) j++;
        const int64_t pos1 = (int64_t)p + 1;
        if (!(small_event && p >= st && p < en)) { // :253-256
            for (int k = 0; k < nw; k++) {
                if (pos1 < w[k][0] || pos1 > w[k][1]) continue;
                for (int64_t s = i; s < j; s++) {
                    const uint32_t cl = cls[s];
                    if (!cl) continue;
                    uint32_t ka = 0;
                    bool is_c;
                    if (whole) {
                        if (vt == UZ_VT_DEL) ka = (cl >> UZ_CL_DEL_SHIFT) & 3;
                        else if (vt == UZ_VT_DUP) ka = (cl >> UZ_CL_DUP_SHIFT) & 3;
                        is_c = ka != 0;
                    } else is_c = (cl & UZ_CL_CAND) != 0;
                    const bool is_h = (cl & UZ_CL_HET) != 0;
                    for (int r = 0; r < mult; r++) {
                        if (is_h) {
                            if (FILL) het_idx[oh + nh] = (int32_t)s;
                            nh++;
                        }
                        if (is_c) {
                            if (FILL) {
                                cand_idx[oc + nc] = (int32_t)s;
                                cand_flags[oc + nc] = (uint8_t)(((cl & UZ_CL_ALT_DAD) ? UZ_CF_ALT_DAD : 0) | (ka << UZ_CF_KA_SHIFT));
                            }
                            nc++;
                        }
                    }
                }
            }
        }
        i = j;
    }
}

// Breakpoint / point mode: one WAVE per DNM.  A window is a contiguous index range of the sites table (two lower bounds), walked
// 64 sites per round with the list positions from ballots; a DNM with two windows (an event longer than search_dist) has
// them one after the other -- window 1 lies wholly before window 2 unless they overlap, and then the reference's order
// (window-1 copy before window-2 copy of the same position) is kept by lane 0 walking the sites alone.
template <bool FILL>
__global__ __launch_bounds__(256) void k_window_wave(WinArgs a, int32_t *cnt_c, int32_t *cnt_h, const int64_t *off_c, const int64_t *off_h,
                                                     int32_t *cand_idx, uint8_t *cand_flags, int32_t *het_idx) {
    const int32_t d = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (d >= a.n) return;
    if (FILL && (off_c[d + 1] > a.cap_c || off_h[d + 1] > a.cap_h)) return;
    int64_t nc = 0, nh = 0;
    const int32_t c = a.contig[d];
    const uint8_t *__restrict__ cls = a.fam_idx ? a.cls_of[a.fam_idx[d]] : a.cls;
    if (c >= 0 && c < a.n_contigs) {
        const int64_t clo = a.contig_off[c], chi = a.contig_off[c + 1];
        const int64_t st = a.start[d], en = a.end[d];
        int64_t w[2][2];
        int nw = 1;
        w[0][0] = st - a.sd; w[0][1] = st + a.sd; // 1-based POS window built from the 0-based start (:24-31)
        if ((a.mode & UZ_FIND_SECOND_WINDOW) && (en - st) > a.sd) { w[1][0] = en - a.sd; w[1][1] = en + a.sd; nw = 2; }
        for (int k = 0; k < nw; k++) if (w[k][0] < 1) w[k][0] = 1;
        const int64_t oc = FILL ? off_c[d] : 0, oh = FILL ? off_h[d] : 0;
        if (nw == 2 && w[1][0] <= w[0][1]) { // overlapping windows: a site can be listed twice, in the reference's order
            if (lane == 0) window_serial<FILL>(a, d, cls, clo, chi, w, nw, oc, oh, cand_idx, cand_flags, het_idx, nc, nh);
        } else {
            const int mult = a.mult[d];
            const bool small_event = (en - st) < 20;
            const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
            for (int k = 0; k < nw; k++) {
                // (count pass: the window's first site by a 64-ary search, its end by walking -- a window holds about a round of sites, and a second
                // search was four more dependent loads in front of the walk: the pass is bound by its chain, 100 k waves in ~12 rounds of the chip)
                int64_t lo, hi;
                if (FILL) { lo = a.range[4 * (int64_t)d + 2 * k]; hi = a.range[4 * (int64_t)d + 2 * k + 1]; }
                else { lo = lower_bound_wave(a.pos, clo, chi, w[k][0] - 1, lane); hi = chi; }
                for (int64_t b = lo; b < hi; b += 64) {
                    const int64_t sidx = b + lane;
                    bool is_c = false, is_h = false;
                    uint32_t cl = 0;
                    bool in = sidx < hi;
                    if (in) {
                        cl = cls[sidx];
                        const int32_t p = a.pos[sidx];
                        if (!FILL) in = (int64_t)p < w[k][1]; // (as lower_bound(pos, w[k][1]): the sites below it)
                        if (in && cl && !(small_event && p >= st && p < en)) { // :253-256
                            is_c = (cl & UZ_CL_CAND) != 0;
                            is_h = (cl & UZ_CL_HET) != 0;
                        }
                    }
                    if (!FILL) {
                        const int n_in = __popcll(__ballot(in)); // (sorted: a prefix of the lanes)
                        if (n_in < 64) hi = b + n_in;            // the window ends in this round
                    }
                    const unsigned long long bc = __ballot(is_c), bh = __ballot(is_h);
                    if (FILL) {
                        if (is_h) for (int r = 0; r < mult; r++) het_idx[oh + (nh + __popcll(bh & below)) * mult + r] = (int32_t)sidx;
                        if (is_c) {
                            const uint8_t fl = (uint8_t)((cl & UZ_CL_ALT_DAD) ? UZ_CF_ALT_DAD : 0);
                            for (int r = 0; r < mult; r++) {
                                const int64_t at = oc + (nc + __popcll(bc & below)) * mult + r;
                                cand_idx[at] = (int32_t)sidx;
                                cand_flags[at] = fl;
                            }
                        }
                    }
                    nc += __popcll(bc); nh += __popcll(bh);
                }
                if (!FILL && lane == 0) { a.range[4 * (int64_t)d + 2 * k] = lo; a.range[4 * (int64_t)d + 2 * k + 1] = hi; }
            }
            nc *= mult; nh *= mult;
        }
    }
    if (!FILL && lane == 0) { cnt_c[d] = (int32_t)nc; cnt_h[d] = (int32_t)nh; }
}

// Whole-region mode (CNV interior, sv_phaser.py:375-389) visits hundreds to thousands of sites per event: one WAVE per
// DNM, 64 consecutive sites per round, list positions from ballots.  One window, so the visiting order is the table
// order and every site of the index range lies inside the window (the range comes from the two lower bounds).
template <bool FILL>
__global__ __launch_bounds__(256) void k_window_region(WinArgs a, int32_t *cnt_c, int32_t *cnt_h, const int64_t *off_c, const int64_t *off_h,
                                                       int32_t *cand_idx, uint8_t *cand_flags, int32_t *het_idx) {
    const int32_t d = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (d >= a.n) return;
    if (FILL && (off_c[d + 1] > a.cap_c || off_h[d + 1] > a.cap_h)) return;
    int64_t nc = 0, nh = 0;
    const int32_t c = a.contig[d];
    const uint8_t *__restrict__ cls = a.fam_idx ? a.cls_of[a.fam_idx[d]] : a.cls;
    if (c >= 0 && c < a.n_contigs) {
        const int64_t clo = a.contig_off[c], chi = a.contig_off[c + 1];
        const int64_t st = a.start[d], en = a.end[d];
        int64_t w0 = st - a.sd;
        const int64_t w1 = en + a.sd;
        if (w0 < 1) w0 = 1;
        const int vt = a.vartype[d];
        const int mult = a.mult[d];
        const bool small_event = (en - st) < 20;
        int64_t lo, hi;
        if (FILL) { lo = a.range[4 * (int64_t)d]; hi = a.range[4 * (int64_t)d + 1]; }
        else {
            lo = lower_bound_wave(a.pos, clo, chi, w0 - 1, lane);
            hi = lower_bound_wave(a.pos, lo, chi, w1, lane);
            if (lane == 0) { a.range[4 * (int64_t)d] = lo; a.range[4 * (int64_t)d + 1] = hi; }
        }
        const int64_t oc = FILL ? off_c[d] : 0, oh = FILL ? off_h[d] : 0;
        const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
        for (int64_t b = lo; b < hi; b += 64) {
            const int64_t sidx = b + lane;
            bool is_c = false, is_h = false;
            uint32_t cl = 0, ka = 0;
            if (sidx < hi) {
                cl = cls[sidx];
                const int32_t p = a.pos[sidx];
                if (cl && !(small_event && p >= st && p < en)) { // :253-256
                    if (vt == UZ_VT_DEL) ka = (cl >> UZ_CL_DEL_SHIFT) & 3;
                    else if (vt == UZ_VT_DUP) ka = (cl >> UZ_CL_DUP_SHIFT) & 3;
                    is_c = ka != 0;
                    is_h = (cl & UZ_CL_HET) != 0;
                }
            }
            const unsigned long long bc = __ballot(is_c), bh = __ballot(is_h);
            if (FILL) {
                if (is_h) for (int r = 0; r < mult; r++) het_idx[oh + (nh + __popcll(bh & below)) * mult + r] = (int32_t)sidx;
                if (is_c) {
                    const uint8_t fl = (uint8_t)(((cl & UZ_CL_ALT_DAD) ? UZ_CF_ALT_DAD : 0) | (ka << UZ_CF_KA_SHIFT));
                    for (int r = 0; r < mult; r++) {
                        const int64_t at = oc + (nc + __popcll(bc & below)) * mult + r;
                        cand_idx[at] = (int32_t)sidx;
                        cand_flags[at] = fl;
                    }
                }
            }
            nc += __popcll(bc); nh += __popcll(bh);
        }
        nc *= mult; nh *= mult;
    }
    if (!FILL && lane == 0) { cnt_c[d] = (int32_t)nc; cnt_h[d] = (int32_t)nh; }
}

// exclusive scan of two count arrays into int64 offsets (n+1 entries), a tile of 4096 counts per workgroup: the tiles' sums first
// (k_scan2_sums), then every workgroup adds up the sums of the tiles before its own and scans its tile (coalesced 16-byte loads,
// wave-shuffle scan).  (Through round 6 ONE workgroup walked the arrays tile by tile: 88 us for the 100 k DNMs of the bench batch,
// 3.5 us per tile of load latency and two barriers with the rest of the chip idle.)
__device__ __forceinline__ void scan2_load(int32_t n, int32_t i0, const int32_t *c0, const int32_t *c1, int (&v0)[4], int (&v1)[4]) {
    v0[0] = v0[1] = v0[2] = v0[3] = 0; v1[0] = v1[1] = v1[2] = v1[3] = 0;
    if (i0 + 3 < n) {
        const int4 a = *reinterpret_cast<const int4 *>(c0 + i0), b = *reinterpret_cast<const int4 *>(c1 + i0);
        v0[0] = a.x; v0[1] = a.y; v0[2] = a.z; v0[3] = a.w;
        v1[0] = b.x; v1[1] = b.y; v1[2] = b.z; v1[3] = b.w;
    } else {
        for (int k = 0; k < 4; k++) if (i0 + k < n) { v0[k] = c0[i0 + k]; v1[k] = c1[i0 + k]; }
    }
}
__global__ __launch_bounds__(1024) void k_scan2_sums(int32_t n, const int32_t *c0, const int32_t *c1, int64_t *part /* [2 tiles] */) {
    __shared__ long long wsum[2][16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int v0[4], v1[4];
    scan2_load(n, (int32_t)blockIdx.x * 4096 + 4 * t, c0, c1, v0, v1);
    long long s0 = (long long)v0[0] + v0[1] + v0[2] + v0[3], s1 = (long long)v1[0] + v1[1] + v1[2] + v1[3];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
    if (lane == 0) { wsum[0][wv] = s0; wsum[1][wv] = s1; }
    __syncthreads();
    if (t < 2) {
        long long a = 0;
        for (int w = 0; w < 16; w++) a += wsum[t][w];
        part[2 * (size_t)blockIdx.x + t] = a;
    }
}
__global__ __launch_bounds__(1024) void k_scan2(int32_t n, const int32_t *c0, const int32_t *c1, const int64_t *__restrict__ part, int64_t *o0, int64_t *o1) {
    __shared__ int wsum[2][16];
    __shared__ long long csum[2][16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int32_t i0 = (int32_t)blockIdx.x * 4096 + 4 * t;
    int v0[4], v1[4];
    scan2_load(n, i0, c0, c1, v0, v1);
    long long q0 = 0, q1 = 0; // the tiles before this one
    for (int b = t; b < (int)blockIdx.x; b += 1024) { q0 += part[2 * (size_t)b]; q1 += part[2 * (size_t)b + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { q0 += __shfl_xor(q0, off, 64); q1 += __shfl_xor(q1, off, 64); }
    const int s0 = v0[0] + v0[1] + v0[2] + v0[3], s1 = v1[0] + v1[1] + v1[2] + v1[3];
    int in0 = s0, in1 = s1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u0 = __shfl_up(in0, off, 64), u1 = __shfl_up(in1, off, 64);
        if (lane >= off) { in0 += u0; in1 += u1; }
    }
    if (lane == 63) { wsum[0][wv] = in0; wsum[1][wv] = in1; }
    if (lane == 0) { csum[0][wv] = q0; csum[1][wv] = q1; }
    __syncthreads();
    int64_t base0 = 0, base1 = 0, pre0 = 0, pre1 = 0, tot0 = 0, tot1 = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) {
        const int a = wsum[0][w], b = wsum[1][w];
        if (w < wv) { pre0 += a; pre1 += b; }
        tot0 += a; tot1 += b;
        base0 += csum[0][w]; base1 += csum[1][w];
    }
    int64_t p0 = base0 + pre0 + (in0 - s0), p1 = base1 + pre1 + (in1 - s1);
    for (int k = 0; k < 4; k++) {
        if (i0 + k < n) { o0[i0 + k] = p0; o1[i0 + k] = p1; }
        p0 += v0[k]; p1 += v1[k];
    }
    if (t == 0 && blockIdx.x == gridDim.x - 1) { o0[n] = base0 + tot0; o1[n] = base1 + tot1; }
}

// K6: allele-balance phasing of a DEL / DUP from the whole-region candidate list (phase_by_snvs, sv_phaser.py:71-85:
// every candidate votes for the parent its kid_allele names) and the decision of summarize_record (unfazed.py:193-298:
// the read-backed branch on the counts handed in, then the CNV branch with its AMBIGUOUS_BOTH / ambiguous merges).
// One wave per DNM; the ordered position lists come from ballots (dad's sites first, then mom's, inside the DNM's
// slice of the candidate list).
__global__ __launch_bounds__(256) void k_cnv_count(int32_t n, const int64_t *__restrict__ cand_off, const int32_t *__restrict__ cand_idx,
                                                   const uint8_t *__restrict__ cand_flags, const int32_t *__restrict__ spos,
                                                   const uint8_t *__restrict__ vartype, const int32_t *__restrict__ rb_counts, int ratio,
                                                   int32_t *cnv_counts, int32_t *cnv_pos, int32_t *origin, int32_t *evidence, int32_t *etype) {
    const int32_t d = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (d >= n) return;
    const int64_t c0 = cand_off[d], c1 = cand_off[d + 1];
    const int vt = vartype[d];
    const bool cnv = vt == UZ_VT_DEL || vt == UZ_VT_DUP; // sv_phaser.py:401
    int n_dad = 0, n_mom = 0;
    if (cnv) {
        for (int64_t b = c0; b < c1; b += 64) { // count
            const int64_t i = b + lane;
            bool dad = false, mom = false;
            if (i < c1) {
                const uint32_t fl = cand_flags[i];
                const uint32_t ka = (fl >> UZ_CF_KA_SHIFT) & 3u;
                const bool alt_dad = (fl & UZ_CF_ALT_DAD) != 0;
                const bool is_dad = (ka == UZ_KA_ALT_PARENT) == alt_dad; // site[site["kid_allele"]] is the dad
                dad = ka != 0 && is_dad; mom = ka != 0 && !is_dad;
            }
            n_dad += __popcll(__ballot(dad)); n_mom += __popcll(__ballot(mom));
        }
        int o_dad = 0, o_mom = 0;
        for (int64_t b = c0; b < c1; b += 64) { // ordered fill
            const int64_t i = b + lane;
            bool dad = false, mom = false;
            int32_t p = 0;
            if (i < c1) {
                const uint32_t fl = cand_flags[i];
                const uint32_t ka = (fl >> UZ_CF_KA_SHIFT) & 3u;
                const bool alt_dad = (fl & UZ_CF_ALT_DAD) != 0;
                const bool is_dad = (ka == UZ_KA_ALT_PARENT) == alt_dad;
                dad = ka != 0 && is_dad; mom = ka != 0 && !is_dad;
                p = spos[cand_idx[i]];
            }
            const unsigned long long bd = __ballot(dad), bm = __ballot(mom);
            const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
            if (dad) cnv_pos[c0 + o_dad + __popcll(bd & below)] = p;
            if (mom) cnv_pos[c0 + n_dad + o_mom + __popcll(bm & below)] = p;
            o_dad += __popcll(bd); o_mom += __popcll(bm);
        }
    }
    if (lane != 0) return;
    cnv_counts[2 * d] = n_dad; cnv_counts[2 * d + 1] = n_mom;
    // summarize_record: 1 dad, 2 mom, 3 "dad|mom", 0 None
    long long dr = 0, mr = 0, ds = 0, ms = 0;
    if (rb_counts) { dr = rb_counts[4 * d]; mr = rb_counts[4 * d + 1]; ds = rb_counts[4 * d + 2]; ms = rb_counts[4 * d + 3]; }
    const long long r = ratio;
    int org = UZ_OR_NONE, et = 0;
    long long ev = 0;
    bool ambig = false;
    if (dr > 0 && dr >= r * mr) { org = UZ_OR_DAD; ev = ds; et = UZ_ET_READBACKED; }
    else if (mr > 0 && mr >= r * dr) { org = UZ_OR_MOM; ev = ms; et = UZ_ET_READBACKED; }
    else if (dr > 0 && mr > 0) { org = UZ_OR_AMBIGUOUS; ev = dr + mr; et = UZ_ET_AMBIGUOUS_READBACKED; ambig = true; }
    const long long cd = n_dad, cm = n_mom;
    if (cd > 0 && cd >= r * cm) {
        if (org == UZ_OR_MOM && !(et & UZ_ET_READBACKED)) { // unreachable with these branches (kept as written, :241-248)
            org = UZ_OR_NONE; ev += cd + cm; et = UZ_ET_AMBIGUOUS_BOTH; ambig = true;
        } else {
            org = UZ_OR_DAD; ev = cd;
            if (et & UZ_ET_AMBIGUOUS_READBACKED) { et &= ~UZ_ET_AMBIGUOUS_READBACKED; ambig = false; }
            et |= UZ_ET_ALLELE_BALANCE;
        }
    } else if (cm > 0 && cm >= r * cd) {
        if (org == UZ_OR_DAD && !(et & UZ_ET_READBACKED)) {
            org = UZ_OR_NONE; ev += cd + cm; et = UZ_ET_AMBIGUOUS_BOTH; ambig = true;
        } else {
            org = UZ_OR_MOM; ev = cm;
            if (et & UZ_ET_AMBIGUOUS_READBACKED) et &= ~UZ_ET_AMBIGUOUS_READBACKED; // `ambig` stays set: the reference forgets to clear it here (:277-278)
            et |= UZ_ET_ALLELE_BALANCE;
        }
    } else if (cd + cm > 0 && !(et & UZ_ET_READBACKED)) {
        org = UZ_OR_NONE; ev += cd + cm; et |= UZ_ET_AMBIGUOUS_ALLELE_BALANCE; ambig = true;
    }
    if (ambig) et |= UZ_ET_AMBIG_FLAG;
    origin[d] = org; evidence[d] = (int32_t)ev; etype[d] = et;
}

SiteParams make_site_params(const uz_params &p) {
    SiteParams s;
    s.min_gt_qual = p.min_gt_qual;
    s.min_depth = p.min_depth;
    s.lo_ref = p.ab_homref[0]; s.hi_ref = p.ab_homref[1];
    s.lo_alt = p.ab_homalt[0]; s.hi_alt = p.ab_homalt[1];
    s.lo_het = p.ab_het[0]; s.hi_het = p.ab_het[1];
    return s;
}

bool site_params_equal(const uz_params &a, const uz_params &b) {
    return a.min_gt_qual == b.min_gt_qual && a.min_depth == b.min_depth &&
           memcmp(a.ab_homref, b.ab_homref, sizeof(a.ab_homref)) == 0 &&
           memcmp(a.ab_homalt, b.ab_homalt, sizeof(a.ab_homalt)) == 0 &&
           memcmp(a.ab_het, b.ab_het, sizeof(a.ab_het)) == 0;
}

} // namespace

__global__ void k_fold_complex(uint8_t *gt, const uint8_t *sflags, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gt[i] = (uint8_t)((gt[i] & 0x3Fu) | ((sflags[i] & UZ_SF_COMPLEX) ? 0x40u : 0u));
}

// the nine eight-bit columns of the link form (uz_family_view.ref_depth8 ...) into the 16-bit ones the kernels read.  `missing`: the
// byte that stands for a missing value (254 in a depth column -- 255 there is "see the wide list": any value will do, the site's class is
// rewritten from the list -- 255 in a quality column)
struct Widen8 { const uint8_t *s[9]; uint16_t *d[9]; };
__global__ __launch_bounds__(256) void k_widen8(int64_t n, Widen8 w) { // (one launch for the nine columns: a chunk of the staged pass makes ~40 launches, each waits for room beside the read stage)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = w.s[k][i];
#pragma unroll
    for (int k = 0; k < 9; k++) w.d[k][i] = (uint16_t)(v[k] == (k < 6 ? (uint32_t)UZ_U8_MISSING : 255u) ? 0xFFFFu : v[k]);
}
void uz_family_widen(uz_ctx *c, FamilyDev &f, int64_t n) {
    if (!f.widen_pending) return;
    f.widen_pending = false;
    if (n <= 0) return;
    const unsigned nb = (unsigned)((n + 255) / 256);
    Widen8 w;
    for (int m = 0; m < 3; m++) {
        w.s[m] = f.stage8[m]; w.s[3 + m] = f.stage8[3 + m]; w.s[6 + m] = f.stage8[6 + m];
        w.d[m] = f.rd[m]; w.d[3 + m] = f.ad[m]; w.d[6 + m] = f.gq[m];
    }
    hipLaunchKernelGGL(k_widen8, dim3(nb), dim3(256), 0, c->stream, n, w);
    UZ_HIP(hipGetLastError());
}

void uz_fold_complex(uz_ctx *c, uint8_t *gt, const uint8_t *sflags, int64_t n) {
    if (n <= 0) return;
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_fold_complex, dim3((unsigned)nb), dim3(256), 0, c->stream, gt, sflags, n);
    UZ_HIP(hipGetLastError());
}

bool uz_site_scan_fresh(const uz_ctx *c, const FamilyDev &f, bool need_cnv) {
    return f.cls_valid && (f.cls_has_cnv || !need_cnv) && site_params_equal(f.cls_params, c->P);
}

#ifndef UZ_SITE_SPT
#define UZ_SITE_SPT 8
#endif

namespace {

void ensure_ab_lut(uz_ctx *c, const SiteParams &sp) {
    if (c->ab_lut_valid && site_params_equal(c->ab_lut_params, c->P)) return;
    c->ab_lut.ensure((size_t)4 * UZ_AB_LUT_N * 2 + 4 + 4 * UZ_AB_LDS_T);
    int *flag = c->ab_lut.p + (size_t)4 * UZ_AB_LUT_N * 2;
    UZ_HIP(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(k_build_ab_lut, dim3((4 * UZ_AB_LUT_N + 255) / 256), dim3(256), 0, c->stream, sp, (int2 *)c->ab_lut.p,
                       (uint32_t *)(flag + 4), flag);
    UZ_HIP(hipGetLastError());
    int special = 0;
    UZ_HIP(hipMemcpyAsync(&special, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    UZ_HIP(hipStreamSynchronize(c->stream));
    c->ab_lut_t0_special = special != 0;
    c->ab_lut_valid = true;
    c->ab_lut_params = c->P;
}

FamPtrs fam_ptrs(const FamilyDev &f) {
    FamPtrs fp;
    fp.gt = f.gt;
    for (int m = 0; m < 3; m++) { fp.rd[m] = f.rd[m]; fp.ad[m] = f.ad[m]; fp.gq[m] = f.gq[m]; }
    return fp;
}

// one family (batch == nullptr) or n_fam families of the same sites table
void launch_site_scan(uz_ctx *c, const FamPtrs &fp, uint8_t *cls, int64_t n, bool with_cnv, const FamBatchItem *batch, int n_fam) {
    const SiteParams sp = make_site_params(c->P);
    ensure_ab_lut(c, sp);
    static const int spt = [] { const char *e = getenv("UZ_SITE_SPT"); return e ? atoi(e) : UZ_SITE_SPT; }();
    ProfScope ps(c, UZ_K_SITE_SCAN);
    auto launch = [&](auto kern, int SPT) {
        const int64_t n_chunks = (n + 256 * SPT - 1) / (256 * SPT);
        static const int wgs = [] { const char *e = getenv("UZ_SITE_WGS"); return e ? atoi(e) : 4096; }();
        int64_t nb = n_chunks < wgs ? n_chunks : wgs; // 256 CUs x 8 resident workgroups, grid-stride over the chunks
        if (batch && n_fam > 1) { // the families fill the chip together
            const int64_t per = (wgs + n_fam - 1) / n_fam;
            nb = std::max<int64_t>(1, std::min<int64_t>(n_chunks, std::max<int64_t>(per, 16)));
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nb, (unsigned)(batch ? n_fam : 1)), dim3(256), 0, c->stream, fp, cls, n, sp,
                           (const int2 *)c->ab_lut.p, (const uint32_t *)(c->ab_lut.p + (size_t)4 * UZ_AB_LUT_N * 2 + 4), batch);
    };
    const bool t0 = c->ab_lut_t0_special;
#define UZ_K1_PICK(SPTV, BATCHV)                                                                                              \
    do {                                                                                                                        \
        if (with_cnv) { if (t0) launch(k_site_scan<SPTV, true, true, BATCHV>, SPTV); else launch(k_site_scan<SPTV, true, false, BATCHV>, SPTV); } \
        else { if (t0) launch(k_site_scan<SPTV, false, true, BATCHV>, SPTV); else launch(k_site_scan<SPTV, false, false, BATCHV>, SPTV); }        \
    } while (0)
    if (batch) { if (spt == 16) UZ_K1_PICK(16, true); else UZ_K1_PICK(8, true); }
    else { if (spt == 16) UZ_K1_PICK(16, false); else UZ_K1_PICK(8, false); }
#undef UZ_K1_PICK
    UZ_HIP(hipGetLastError());
}

} // namespace

static void launch_site_scan_wide(uz_ctx *c, FamilyDev &f, bool with_cnv) {
    if (f.n_wide <= 0) return;
    const SiteParams P = make_site_params(c->P);
    const unsigned nb = (unsigned)((f.n_wide + 255) / 256);
    if (with_cnv) hipLaunchKernelGGL(k_site_scan_wide<true>, dim3(nb), dim3(256), 0, c->stream, fam_ptrs(f), f.cls, f.n_wide, f.wide_site, f.wide_depth, P);
    else hipLaunchKernelGGL(k_site_scan_wide<false>, dim3(nb), dim3(256), 0, c->stream, fam_ptrs(f), f.cls, f.n_wide, f.wide_site, f.wide_depth, P);
    UZ_HIP(hipGetLastError());
}

static void check_gq_clamp(const uz_ctx *c, const FamilyDev &f) {
    UZ_REQUIRE(!f.gq_clamped || c->P.min_gt_qual <= 254, UZ_E_STATE,
               "this family was staged with eight-bit genotype qualities (clamped at 254): --min-gt-qual above 254 needs the 16-bit columns");
}
void uz_launch_site_scan(uz_ctx *c, FamilyDev &f, const SitesDev &s, bool with_cnv) {
    check_gq_clamp(c, f);
    if (s.n > 0) { launch_site_scan(c, fam_ptrs(f), f.cls, s.n, with_cnv, nullptr, 1); launch_site_scan_wide(c, f, with_cnv); }
    f.cls_has_cnv = with_cnv;
    f.cls_valid = true;
    f.cls_params = c->P;
}

void uz_launch_site_scan_many(uz_ctx *c, FamilyDev *const *fams, int n_fam, const SitesDev &s, bool with_cnv) {
    if (n_fam <= 0) return;
    for (int k = 0; k < n_fam; k++) check_gq_clamp(c, *fams[k]);
    if (s.n > 0) {
        std::vector<FamBatchItem> items((size_t)n_fam);
        for (int k = 0; k < n_fam; k++) { items[(size_t)k].f = fam_ptrs(*fams[k]); items[(size_t)k].cls = fams[k]->cls; }
        c->fam_batch.ensure((size_t)n_fam * sizeof(FamBatchItem));
        UZ_HIP(hipMemcpyAsync(c->fam_batch.p, items.data(), items.size() * sizeof(FamBatchItem), hipMemcpyHostToDevice, c->stream));
        UZ_HIP(hipStreamSynchronize(c->stream)); // `items` is pageable host memory
        launch_site_scan(c, items[0].f, items[0].cls, s.n, with_cnv, (const FamBatchItem *)c->fam_batch.p, n_fam);
        for (int k = 0; k < n_fam; k++) launch_site_scan_wide(c, *fams[k], with_cnv);
    }
    for (int k = 0; k < n_fam; k++) {
        fams[k]->cls_has_cnv = with_cnv;
        fams[k]->cls_valid = true;
        fams[k]->cls_params = c->P;
    }
}

void uz_launch_find(uz_ctx *c, FamilyDev &f, const SitesDev &s, int mode, bool host_offsets) {
    const int32_t n = c->dn.n;
    c->cnt_c.ensure((size_t)n + 1);
    c->cnt_h.ensure((size_t)n + 1);
    c->cand_off.ensure((size_t)n + 1);
    c->het_off.ensure((size_t)n + 1);
    if (host_offsets || n <= 0) { c->cand_off_h.assign((size_t)n + 1, 0); c->het_off_h.assign((size_t)n + 1, 0); }
    else { c->cand_off_h.resize((size_t)n + 1); c->het_off_h.resize((size_t)n + 1); } // only the totals are read back
    c->n_cand = c->n_het = 0;
    if (n > 0) {
        WinArgs a;
        a.n = n;
        a.contig = c->dn.contig.p; a.start = c->dn.start.p; a.end = c->dn.end.p;
        a.vartype = c->dn.vartype.p; a.mult = c->dn.mult.p;
        a.contig_off = s.contig_off; a.n_contigs = s.n_contigs;
        a.pos = s.pos; a.cls = f.cls;
        a.cls_of = c->cohort_on ? (const uint8_t *const *)c->fam_cls.p : nullptr;
        a.fam_idx = c->cohort_on ? c->dn_fam.p : nullptr;
        a.sd = c->P.search_dist;
        a.mode = mode;
        c->win_range.ensure((size_t)4 * n + 4);
        a.range = c->win_range.p;
        const unsigned nbw = (unsigned)(((int64_t)n * 64 + 255) / 256); // one wave per DNM
        {
            ProfScope ps(c, UZ_K_WINDOW_COUNT);
            if (mode & UZ_FIND_WHOLE_REGION)
                hipLaunchKernelGGL(k_window_region<false>, dim3(nbw), dim3(256), 0, c->stream, a, c->cnt_c.p, c->cnt_h.p,
                                   (const int64_t *)nullptr, (const int64_t *)nullptr, (int32_t *)nullptr, (uint8_t *)nullptr,
                                   (int32_t *)nullptr);
            else
                hipLaunchKernelGGL(k_window_wave<false>, dim3(nbw), dim3(256), 0, c->stream, a, c->cnt_c.p, c->cnt_h.p,
                                   (const int64_t *)nullptr, (const int64_t *)nullptr, (int32_t *)nullptr,
                                   (uint8_t *)nullptr, (int32_t *)nullptr);
            UZ_HIP(hipGetLastError());
            const unsigned tiles = (unsigned)(((int64_t)n + 4095) / 4096);
            c->scan_part.ensure((size_t)2 * tiles);
            hipLaunchKernelGGL(k_scan2_sums, dim3(tiles), dim3(1024), 0, c->stream, n, c->cnt_c.p, c->cnt_h.p, c->scan_part.p);
            UZ_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_scan2, dim3(tiles), dim3(1024), 0, c->stream, n, c->cnt_c.p, c->cnt_h.p, (const int64_t *)c->scan_part.p, c->cand_off.p,
                               c->het_off.p);
            UZ_HIP(hipGetLastError());
        }
        auto fill = [&](int64_t cap_c, int64_t cap_h) {
            a.cap_c = cap_c; a.cap_h = cap_h;
            ProfScope ps(c, UZ_K_WINDOW_FILL);
            if (mode & UZ_FIND_WHOLE_REGION)
                hipLaunchKernelGGL(k_window_region<true>, dim3(nbw), dim3(256), 0, c->stream, a, (int32_t *)nullptr, (int32_t *)nullptr,
                                   (const int64_t *)c->cand_off.p, (const int64_t *)c->het_off.p, c->cand_idx.p, c->cand_flags.p,
                                   c->het_idx.p);
            else
                hipLaunchKernelGGL(k_window_wave<true>, dim3(nbw), dim3(256), 0, c->stream, a, (int32_t *)nullptr,
                                   (int32_t *)nullptr, (const int64_t *)c->cand_off.p, (const int64_t *)c->het_off.p,
                                   c->cand_idx.p, c->cand_flags.p, c->het_idx.p);
            UZ_HIP(hipGetLastError());
        };
        // The lists are sized by the totals, and the totals are on the device: waiting for them in front of the fill pass is a host
        // round trip with an idle device (two per chunk of a staged pass, one more in the allele-balance stage).  So the fill pass
        // first runs on the lists AS THEY ARE -- a set of lists keeps the room of the batches it held before -- with the room as a
        // bound (a DNM whose slice would end beyond it writes nothing), the totals come back behind it, and only a batch that
        // outgrew the room is filled again after the lists have grown.  (UZ_TEST_FIND_CAP: test hook, a room too small.)
        int64_t room_c = (int64_t)std::min(c->cand_idx.cap, c->cand_flags.cap) - 1, room_h = (int64_t)c->het_idx.cap - 1;
        if (const char *e = getenv("UZ_TEST_FIND_CAP")) { room_c = std::min<int64_t>(room_c, atoll(e)); room_h = std::min<int64_t>(room_h, atoll(e)); }
        const bool tried = room_c > 0 && room_h > 0;
        if (tried) fill(room_c, room_h);
        if (host_offsets) {
            // By a copy kernel into page-locked memory, not by hipMemcpyAsync: the DMA engine is in order, and inside a staged pass these two small
            // copies would wait behind the records of the chunk before (65 MB: 1.4 ms per chunk of the 100 k pass, round 5's trace) -- the host
            // sat in uz_find while the link idled, and enqueued the next chunk's records only afterwards
            const size_t ob = ((size_t)n + 1) * sizeof(int64_t), obr = (ob + 255) & ~(size_t)255;
            if (c->find_pin_cap < 2 * obr) {
                if (c->find_pin) (void)hipHostFree(c->find_pin);
                c->find_pin = nullptr;
                c->find_pin_cap = 0; // (set again only once the new block exists: a failed allocation must not leave a capacity behind)
                const size_t want = 2 * obr + obr / 2 + 4096;
                uint8_t *pin = nullptr;
                UZ_HIP(hipHostMalloc((void **)&pin, want, hipHostMallocDefault));
                c->find_pin = pin;
                c->find_pin_cap = want;
            }
            uz_kcopy(c, c->find_pin, c->cand_off.p, ob);
            uz_kcopy(c, c->find_pin + obr, c->het_off.p, ob);
            UZ_HIP(hipStreamSynchronize(c->stream));
            memcpy(c->cand_off_h.data(), c->find_pin, ob);
            memcpy(c->het_off_h.data(), c->find_pin + obr, ob);
        } else { // the read stage only needs the two totals to size the lists: into the pinned mailbox, by a kernel
            int64_t *box = reinterpret_cast<int64_t *>(c->hflags + 4);
            uz_kcopy(c, box, c->cand_off.p + n, sizeof(int64_t));
            uz_kcopy(c, box + 1, c->het_off.p + n, sizeof(int64_t));
            UZ_HIP(hipStreamSynchronize(c->stream));
            c->cand_off_h[n] = box[0];
            c->het_off_h[n] = box[1];
        }
        c->n_cand = c->cand_off_h[n];
        c->n_het = c->het_off_h[n];
        if (!tried || c->n_cand > room_c || c->n_het > room_h) {
            c->cand_idx.ensure((size_t)c->n_cand + 1);
            c->cand_flags.ensure((size_t)c->n_cand + 1);
            c->het_idx.ensure((size_t)c->n_het + 1);
            fill(INT64_MAX, INT64_MAX);
        }
    }
    c->find_valid = true;
    c->find_mode = mode;
}

void uz_launch_cnv(uz_ctx *c, const SitesDev &s, const int32_t *rb_counts_dev, int32_t *cnv_counts, int32_t *cnv_pos, int32_t *origin,
                   int32_t *evidence, int32_t *etype) {
    const int32_t n = c->dn.n;
    if (n <= 0) return;
    ProfScope ps(c, UZ_K_CNV);
    hipLaunchKernelGGL(k_cnv_count, dim3((unsigned)(((int64_t)n * 64 + 255) / 256)), dim3(256), 0, c->stream, n, (const int64_t *)c->cand_off.p,
                       (const int32_t *)c->cand_idx.p, (const uint8_t *)c->cand_flags.p, (const int32_t *)s.pos, (const uint8_t *)c->dn.vartype.p,
                       rb_counts_dev, (int)c->P.evidence_min_ratio, cnv_counts, cnv_pos, origin, evidence, etype);
    UZ_HIP(hipGetLastError());
}
