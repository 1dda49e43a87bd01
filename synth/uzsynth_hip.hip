// uzsynth_hip.hip -- GPU build of the benchmark-scale generator (uzsynth.h): one workgroup
// per DNM sorts the block's record keys in LDS and writes every column in place in HBM.
// TEST / BENCH INFRASTRUCTURE (bench.py, -m gpu tests); not part of the product library.
#include <hip/hip_runtime.h>

#include "uzsynth.h"

#define UZS_NT 256
#define UZS_MAXSEG 4096

__global__ __launch_bounds__(UZS_NT) void k_gen_reads(uzs_cfg c, uzs_sites S, uzs_dnms D, int32_t d0, int32_t d1, uzs_out o) {
    __shared__ unsigned long long keys[UZS_MAXSEG];
    __shared__ unsigned short inv[UZS_MAXSEG];
    __shared__ long long win[2];
    const int nseg = 2 * c.n_pairs;
    int N = 1;
    while (N < nseg) N <<= 1;
    for (int32_t d = d0 + blockIdx.x; d < d1; d += gridDim.x) {
        __syncthreads();
        for (int slot = threadIdx.x; slot < N; slot += UZS_NT) {
            unsigned long long k = ~0ULL;
            if (slot < nseg) {
                uzs_seg s;
                uzs_segment(&c, &D, d, slot >> 1, slot & 1, &s);
                k = uzs_key(&c, &D, d, slot, &s);
            }
            keys[slot] = k;
        }
        if (threadIdx.x == 0) {
            int64_t a, b;
            uzs_site_window(&c, &S, &D, d, &a, &b);
            win[0] = a; win[1] = b;
        }
        __syncthreads();
        for (int k = 2; k <= N; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = threadIdx.x; i < N; i += UZS_NT) {
                    const int x = i ^ j;
                    if (x > i) {
                        const unsigned long long u = keys[i], v = keys[x];
                        const bool up = (i & k) == 0;
                        if ((u > v) == up) { keys[i] = v; keys[x] = u; }
                    }
                }
                __syncthreads();
            }
        for (int p = threadIdx.x; p < nseg; p += UZS_NT) inv[keys[p] & 0xFFFF] = (unsigned short)p;
        __syncthreads();
        for (int p = threadIdx.x; p < nseg; p += UZS_NT) {
            const int slot = (int)(keys[p] & 0xFFFF);
            uzs_seg s;
            uzs_segment(&c, &D, d, slot >> 1, slot & 1, &s);
            uzs_write_record(&c, &D, d, d0, slot, p, inv[slot ^ 1], &s, &o);
        }
        // bases: one lane per 16-byte chunk of a row, written as one 16-byte store
        const int chunks = UZS_ROW / 16;
        for (int it = threadIdx.x; it < nseg * chunks; it += UZS_NT) {
            const int p = it / chunks, ch = it % chunks;
            const int slot = (int)(keys[p] & 0xFFFF);
            uzs_seg s;
            uzs_segment(&c, &D, d, slot >> 1, slot & 1, &s);
            uint8_t sq[16], ql[16];
            for (int z = 0; z < 16; z++) { sq[z] = 0; ql[z] = 0; }
            const int i0 = ch * 16;
            int i1 = i0 + 16;
            if (i1 > UZS_READLEN) i1 = UZS_READLEN;
            if (i0 < i1) uzs_fill(&c, &S, &D, d, &s, win[0], win[1], i0, i1, sq, ql);
            const int64_t row = ((int64_t)(d - d0) * nseg + p) * UZS_ROW + i0;
            uint4 a, b;
            memcpy(&a, sq, 16);
            memcpy(&b, ql, 16);
            *reinterpret_cast<uint4 *>(o.seq + row) = a;
            *reinterpret_cast<uint4 *>(o.qual + row) = b;
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
int uzs_gen_reads_hip(const uzs_cfg *c, const uzs_sites *S, const uzs_dnms *D, int32_t d0, int32_t d1, const uzs_out *o) {
    if (2 * c->n_pairs > UZS_MAXSEG) return -2;
    int grid = d1 - d0;
    if (grid > 65535) grid = 65535;
    if (grid <= 0) return 0;
    hipLaunchKernelGGL(k_gen_reads, dim3((unsigned)grid), dim3(UZS_NT), 0, 0, *c, *S, *D, d0, d1, *o);
    if (hipGetLastError() != hipSuccess) return -1;
    return hipDeviceSynchronize() == hipSuccess ? 0 : -1;
}
}
