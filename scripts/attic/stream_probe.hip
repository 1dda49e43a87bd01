// stream_probe.hip -- development aid: what does the K1 access pattern (nine u16 columns + one u8
// column in, one u8 column out, 8 sites per lane) reach with NO arithmetic, next to a plain copy?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Ptrs { const uint16_t *c[9]; const uint8_t *g; };

template <int NS>
__global__ __launch_bounds__(256) void k_streams(Ptrs p, uint8_t *out, int64_t n) {
    const int64_t nch = (n + 2047) / 2048;
    for (int64_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        const int64_t base = (ch * 256 + threadIdx.x) * 8;
        if (base + 8 > n) continue;
        uint4 v[9];
#pragma unroll
        for (int k = 0; k < NS; k++) v[k] = *reinterpret_cast<const uint4 *>(p.c[k] + base);
        uint2 g = *reinterpret_cast<const uint2 *>(p.g + base);
        uint32_t a = g.x, b = g.y;
#pragma unroll
        for (int k = 0; k < NS; k++) { a ^= v[k].x ^ v[k].z; b ^= v[k].y ^ v[k].w; }
        *reinterpret_cast<uint2 *>(out + base) = make_uint2(a, b);
    }
}

__global__ __launch_bounds__(256) void k_copy(const uint4 *in, uint4 *out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}

int main() {
    const int64_t n = 20000000;
    Ptrs p;
    for (int k = 0; k < 9; k++) { void *d; hipMalloc(&d, n * 2 + 64); hipMemset(d, k + 1, n * 2); p.c[k] = (const uint16_t *)d; }
    void *g, *o; hipMalloc(&g, n + 64); hipMemset(g, 1, n); hipMalloc(&o, n + 64); p.g = (const uint8_t *)g;
    void *ci, *co; hipMalloc(&ci, 200000000); hipMalloc(&co, 200000000); hipMemset(ci, 3, 200000000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {2048, 4096, 8192}) {
        float ms;
        for (int rep = 0; rep < 3; rep++) k_streams<9><<<grid, 256>>>(p, (uint8_t *)o, n);
        hipEventRecord(e0); for (int rep = 0; rep < 20; rep++) k_streams<9><<<grid, 256>>>(p, (uint8_t *)o, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("10 streams grid %d: %.1f us -> %.2f TB/s (400 MB)\n", grid, ms / 20 * 1e3, 400e6 / (ms / 20 * 1e-3) / 1e12);
        for (int rep = 0; rep < 3; rep++) k_copy<<<grid, 256>>>((const uint4 *)ci, (uint4 *)co, 200000000 / 16);
        hipEventRecord(e0); for (int rep = 0; rep < 20; rep++) k_copy<<<grid, 256>>>((const uint4 *)ci, (uint4 *)co, 200000000 / 16); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("copy 200MB+200MB grid %d: %.1f us -> %.2f TB/s\n", grid, ms / 20 * 1e3, 400e6 / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
