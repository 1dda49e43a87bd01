// stream_probe2.hip -- development aid: the K1 access pattern COLD (working set far above the
// 256 MB Infinity Cache): ten separate column streams vs the same bytes laid out as one tiled stream
// (per 2048-site tile: the nine u16 columns, then the u8 column), vs a plain copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Ptrs { const uint16_t *c[9]; const uint8_t *g; uint8_t *out; };

__global__ __launch_bounds__(256) void k_streams(const Ptrs *fams, int64_t n) {
    const Ptrs p = fams[blockIdx.y];
    const int64_t nch = (n + 2047) / 2048;
    for (int64_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        const int64_t base = (ch * 256 + threadIdx.x) * 8;
        if (base + 8 > n) continue;
        uint4 v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = *reinterpret_cast<const uint4 *>(p.c[k] + base);
        uint2 g = *reinterpret_cast<const uint2 *>(p.g + base);
        uint32_t a = g.x, b = g.y;
#pragma unroll
        for (int k = 0; k < 9; k++) { a ^= v[k].x ^ v[k].z; b ^= v[k].y ^ v[k].w; }
        *reinterpret_cast<uint2 *>(p.out + base) = make_uint2(a, b);
    }
}

// tile of 2048 sites = 9 * 4096 B of u16 columns + 2048 B of gt = 38912 B
__global__ __launch_bounds__(256) void k_tiled(const uint8_t *const *fam_tiles, uint8_t *const *outs, int64_t n) {
    const uint8_t *t = fam_tiles[blockIdx.y];
    uint8_t *out = outs[blockIdx.y];
    const int64_t nch = (n + 2047) / 2048;
    for (int64_t ch = blockIdx.x; ch < nch; ch += gridDim.x) {
        const uint8_t *tile = t + ch * 38912;
        uint4 v[9];
#pragma unroll
        for (int k = 0; k < 9; k++) v[k] = *reinterpret_cast<const uint4 *>(tile + k * 4096 + threadIdx.x * 16);
        uint2 g = *reinterpret_cast<const uint2 *>(tile + 9 * 4096 + threadIdx.x * 8);
        uint32_t a = g.x, b = g.y;
#pragma unroll
        for (int k = 0; k < 9; k++) { a ^= v[k].x ^ v[k].z; b ^= v[k].y ^ v[k].w; }
        *reinterpret_cast<uint2 *>(out + (ch * 256 + threadIdx.x) * 8) = make_uint2(a, b);
    }
}

__global__ __launch_bounds__(256) void k_copy(const uint4 *in, uint4 *out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) out[i] = in[i];
}

int main() {
    const int64_t n = 20000000 / 2048 * 2048;
    const int F = 8;
    std::vector<Ptrs> hp(F);
    std::vector<uint8_t *> ht(F), ho(F);
    for (int f = 0; f < F; f++) {
        for (int k = 0; k < 9; k++) { void *d; hipMalloc(&d, n * 2 + 64); hipMemset(d, k + 1, n * 2); hp[f].c[k] = (const uint16_t *)d; }
        void *g, *o, *t; hipMalloc(&g, n + 64); hipMemset(g, 1, n); hipMalloc(&o, n + 64); hp[f].g = (const uint8_t *)g; hp[f].out = (uint8_t *)o;
        hipMalloc(&t, n / 2048 * 38912 + 64); hipMemset(t, 2, n / 2048 * 38912); ht[f] = (uint8_t *)t; ho[f] = (uint8_t *)o;
    }
    Ptrs *dp; hipMalloc(&dp, sizeof(Ptrs) * F); hipMemcpy(dp, hp.data(), sizeof(Ptrs) * F, hipMemcpyHostToDevice);
    uint8_t **dt, **dout; hipMalloc(&dt, 8 * F); hipMalloc(&dout, 8 * F);
    hipMemcpy(dt, ht.data(), 8 * F, hipMemcpyHostToDevice); hipMemcpy(dout, ho.data(), 8 * F, hipMemcpyHostToDevice);
    const int64_t cb = (int64_t)F * 190000000; // copy of the same volume
    void *ci, *co; hipMalloc(&ci, cb); hipMalloc(&co, cb / 19); hipMemset(ci, 3, cb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)F * n * 20;
    for (int grid : {512, 1024, 2048}) {
        float ms;
        k_streams<<<dim3(grid, F), 256>>>(dp, n);
        hipEventRecord(e0); for (int rep = 0; rep < 5; rep++) k_streams<<<dim3(grid, F), 256>>>(dp, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("cold 10 streams x %d families, grid %dx%d: %.1f us per family -> %.2f TB/s\n", F, grid, F, ms / 5 / F * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
        k_tiled<<<dim3(grid, F), 256>>>(dt, dout, n);
        hipEventRecord(e0); for (int rep = 0; rep < 5; rep++) k_tiled<<<dim3(grid, F), 256>>>(dt, dout, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("cold tiled layout  x %d families, grid %dx%d: %.1f us per family -> %.2f TB/s\n", F, grid, F, ms / 5 / F * 1e3, bytes / (ms / 5 * 1e-3) / 1e12);
    }
    {
        float ms;
        k_copy<<<8192, 256>>>((const uint4 *)ci, (uint4 *)ci, 0);
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; rep++) k_copy<<<8192, 256>>>((const uint4 *)ci, (uint4 *)ci + cb / 32, cb / 32); // read half, write half
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("cold copy (read %.2f GB + write %.2f GB): %.2f TB/s\n", cb / 2 / 1e9, cb / 2 / 1e9, (double)cb / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
