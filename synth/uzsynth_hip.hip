// uzsynth_hip.hip -- GPU build of the benchmark-scale generator (uzsynth.h): one workgroup per cluster sorts
// the cluster's record keys in LDS and writes every column in place in HBM, directly in the staged format
// (uz_reads_packed_view).  TEST / BENCH INFRASTRUCTURE (bench.py, -m gpu tests); not part of the product library.
#include <hip/hip_runtime.h>

#include "uzsynth.h"

#define UZS_NT 256

// CIGAR words per cluster (the words lie back to back in the table: their offsets are prefix sums)
__global__ __launch_bounds__(UZS_NT) void k_count_ops(uzs_cfg cf, uzs_clusters C, uzs_dnms D, int64_t *ops) {
    __shared__ int part[UZS_NT / 64];
    for (int32_t c = blockIdx.x; c < cf.n_clusters; c += gridDim.x) {
        const int nseg = (int)(2 * (C.pair_off[c + 1] - C.pair_off[c]));
        int mine = 0;
        for (int slot = threadIdx.x; slot < nseg; slot += UZS_NT) {
            uzs_seg s;
            uzs_segment(&cf, &C, &D, c, slot >> 1, slot & 1, &s);
            mine += s.n_ops;
        }
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
        __syncthreads();
        if (threadIdx.x == 0) ops[c] = part[0] + part[1] + part[2] + part[3];
    }
}

// pack 32 bases / qualities (n valid) into 8 + 4 bytes of the staged format (two-bit base rows: A 0, C 1, G 2, T 3, the
// first base of a byte in its top bits; uzs_fill writes nothing else)
__device__ __forceinline__ void pack_unit(const uint8_t *sq, const uint8_t *ql, int n, int thr, uint2 *seq2, uint32_t *qlow) {
    uint32_t w[2] = {0, 0}, qw = 0;
    for (int k = 0; k < n; k++) {
        const uint8_t b = sq[k];
        const uint32_t code = b == 'C' ? 1u : (b == 'G' ? 2u : (b == 'T' ? 3u : 0u));
        const int byte = k >> 2;
        w[byte >> 2] |= (code << (6 - 2 * (k & 3))) << (8 * (byte & 3));
        qw |= (uint32_t)((int)ql[k] < thr) << k;
    }
    *seq2 = make_uint2(w[0], w[1]);
    *qlow = qw;
}

template <int MS>
__global__ __launch_bounds__(UZS_NT) void k_gen_clusters(uzs_cfg cf, uzs_sites S, uzs_dnms D, uzs_clusters C, const int32_t *list, int32_t n_list,
                                                         uzs_out_packed o) {
    __shared__ uint32_t keys[MS];
    __shared__ unsigned short inv[MS];
    __shared__ unsigned short pre[MS];
    __shared__ unsigned char nops[MS];
    __shared__ long long win[2];
    __shared__ int part[UZS_NT + 1];
    for (int32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const int32_t c = list[li];
        const int nseg = (int)(2 * (C.pair_off[c + 1] - C.pair_off[c]));
        int N = 1;
        while (N < nseg) N <<= 1;
        __syncthreads();
        for (int slot = threadIdx.x; slot < N; slot += UZS_NT) {
            uint32_t k = 0xFFFFFFFFu;
            if (slot < nseg) {
                uzs_seg s;
                uzs_segment(&cf, &C, &D, c, slot >> 1, slot & 1, &s);
                k = uzs_key(&C, c, slot, &s);
            }
            keys[slot] = k;
        }
        if (threadIdx.x == 0) {
            int64_t a, b;
            uzs_site_window(&S, &C, c, &a, &b);
            win[0] = a; win[1] = b;
        }
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < N; i += UZS_NT) {
                    const int x = i ^ j;
                    if (x > i) {
                        const uint32_t u = keys[i], v = keys[x];
                        const bool up = (i & k) == 0;
                        if ((u > v) == up) { keys[i] = v; keys[x] = u; }
                    }
                }
                __syncthreads();
            }
        for (int p = threadIdx.x; p < nseg; p += UZS_NT) {
            const int slot = (int)(keys[p] & 0x3FFF);
            inv[slot] = (unsigned short)p;
            uzs_seg s;
            uzs_segment(&cf, &C, &D, c, slot >> 1, slot & 1, &s);
            nops[p] = s.n_ops;
        }
        __syncthreads();
        { // exclusive scan of nops over the sorted order -> pre
            const int chunk = (nseg + UZS_NT - 1) / UZS_NT;
            int lo = threadIdx.x * chunk; if (lo > nseg) lo = nseg;
            int hi = lo + chunk; if (hi > nseg) hi = nseg;
            int sum = 0;
            for (int p = lo; p < hi; p++) sum += nops[p];
            part[threadIdx.x] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {
                int run = 0;
                for (int t = 0; t < UZS_NT; t++) { const int v = part[t]; part[t] = run; run += v; }
            }
            __syncthreads();
            int run = part[threadIdx.x];
            for (int p = lo; p < hi; p++) { pre[p] = (unsigned short)run; run += nops[p]; }
            __syncthreads();
        }
        { // name ids as a decoder hands them out: in order of first appearance (a pair's id = pairs that start before it in the block)
            const int chunk = (nseg + UZS_NT - 1) / UZS_NT;
            int lo = threadIdx.x * chunk; if (lo > nseg) lo = nseg;
            int hi = lo + chunk; if (hi > nseg) hi = nseg;
            int sum = 0;
            for (int p = lo; p < hi; p++) sum += (int)inv[(keys[p] & 0x3FFF) ^ 1] > p;
            part[threadIdx.x] = sum;
            __syncthreads();
            if (threadIdx.x == 0) {
                int run = 0;
                for (int t = 0; t < UZS_NT; t++) { const int v = part[t]; part[t] = run; run += v; }
            }
            __syncthreads();
            int run = part[threadIdx.x];
            for (int p = lo; p < hi; p++) { // (the sort is done: the key's upper bits now hold the rank, its low 14 the slot)
                const uint32_t slot = keys[p] & 0x3FFF;
                keys[p] = slot | ((uint32_t)run << 14);
                run += (int)inv[slot ^ 1] > p;
            }
            __syncthreads();
        }
        const int64_t rec0 = 2 * C.pair_off[c];
        for (int p = threadIdx.x; p < nseg; p += UZS_NT) {
            const int slot = (int)(keys[p] & 0x3FFF);
            uzs_seg s;
            uzs_segment(&cf, &C, &D, c, slot >> 1, slot & 1, &s);
            const int64_t i = rec0 + p;
            o.start[i] = s.start; o.end[i] = s.end; o.tlen[i] = s.tlen;
            const int pm = (int)inv[slot ^ 1];
            o.mate[i] = (int32_t)(rec0 + pm);
            o.qname[i] = (uint32_t)(C.pair_off[c] + (keys[pm > p ? p : pm] >> 14));
            o.flag[i] = s.flag; o.l_seq[i] = (uint16_t)uzs_query_len(&s); o.n_cigar[i] = s.n_ops;
            o.mapq[i] = s.mapq; o.aux[i] = 1; /* mate on the same contig */
            for (int j = 0; j < s.n_ops; j++) o.cigar[C.cigar_off[c] + pre[p] + j] = s.ops[j];
        }
        // bases: one lane per 32-base unit of a row
        for (int it = threadIdx.x; it < nseg * UZS_UNITS; it += UZS_NT) {
            const int p = it / UZS_UNITS, u = it % UZS_UNITS;
            const int slot = (int)(keys[p] & 0x3FFF);
            uzs_seg s;
            uzs_segment(&cf, &C, &D, c, slot >> 1, slot & 1, &s);
            uint8_t sq[32], ql[32];
            const int i0 = u * 32;
            int i1 = i0 + 32;
            if (i1 > UZS_READLEN) i1 = UZS_READLEN;
            uzs_fill(&S, &C, &D, c, &s, win[0], win[1], i0, i1, sq, ql);
            uint2 a;
            uint32_t b;
            pack_unit(sq, ql, i1 - i0, cf.min_base_qual, &a, &b);
            const int64_t unit = (rec0 + p) * UZS_UNITS + u;
            *reinterpret_cast<uint2 *>(o.seq2 + unit * 8) = a;
            *reinterpret_cast<uint32_t *>(o.qlow + unit * 4) = b;
        }
    }
}

extern "C" {

void *uzs_dev_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    return p;
}
void uzs_dev_free(void *p) { (void)hipFree(p); }
int uzs_h2d(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
int uzs_d2h(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
int uzs_set_device(int dev) { return hipSetDevice(dev) == hipSuccess ? 0 : -1; }

// all pointers inside the structs are DEVICE pointers
int uzs_count_ops_hip(const uzs_cfg *cf, const uzs_clusters *C, const uzs_dnms *D, int64_t *ops_dev) {
    if (cf->n_clusters <= 0) return 0;
    int grid = cf->n_clusters > 65535 ? 65535 : cf->n_clusters;
    hipLaunchKernelGGL(k_count_ops, dim3((unsigned)grid), dim3(UZS_NT), 0, 0, *cf, *C, *D, ops_dev);
    if (hipGetLastError() != hipSuccess) return -1;
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}

// list_small / list_big: cluster ids with <= 4096 / more records (device arrays)
int uzs_gen_reads_hip(const uzs_cfg *cf, const uzs_sites *S, const uzs_dnms *D, const uzs_clusters *C, const int32_t *list_small,
                      int32_t n_small, const int32_t *list_big, int32_t n_big, const uzs_out_packed *o) {
    if (n_small > 0) {
        const int grid = n_small > 65535 ? 65535 : n_small;
        hipLaunchKernelGGL(k_gen_clusters<4096>, dim3((unsigned)grid), dim3(UZS_NT), 0, 0, *cf, *S, *D, *C, list_small, n_small, *o);
        if (hipGetLastError() != hipSuccess) return -1;
    }
    if (n_big > 0) {
        const int grid = n_big > 65535 ? 65535 : n_big;
        hipLaunchKernelGGL(k_gen_clusters<UZS_MAXSEG>, dim3((unsigned)grid), dim3(UZS_NT), 0, 0, *cf, *S, *D, *C, list_big, n_big, *o);
        if (hipGetLastError() != hipSuccess) return -1;
    }
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
}
