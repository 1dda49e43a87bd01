// stream_probe3.hip -- development aid: does staggering the bases of the ten column streams (so that they do
// not walk the same HBM channels in lock-step) change the cold streaming rate of the K1 access pattern?
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

int main() {
    const int64_t n = 20000000 / 2048 * 2048;
    const int F = 8;
    for (int64_t stagger : {0LL, 256LL, 4352LL, 69888LL, 1114368LL}) {
        std::vector<Ptrs> hp(F);
        std::vector<void *> bufs;
        for (int f = 0; f < F; f++) {
            const size_t col = (size_t)n * 2 + (1 << 21);
            void *big; hipMalloc(&big, col * 11 + (1 << 22)); hipMemset(big, 1, col * 11);
            bufs.push_back(big);
            uint8_t *b = (uint8_t *)big;
            for (int k = 0; k < 9; k++) hp[f].c[k] = (const uint16_t *)(b + col * k + stagger * k);
            hp[f].g = b + col * 9 + stagger * 9;
            hp[f].out = b + col * 10 + stagger * 10;
        }
        Ptrs *dp; hipMalloc(&dp, sizeof(Ptrs) * F); hipMemcpy(dp, hp.data(), sizeof(Ptrs) * F, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float ms;
        k_streams<<<dim3(1024, F), 256>>>(dp, n);
        hipEventRecord(e0); for (int rep = 0; rep < 5; rep++) k_streams<<<dim3(1024, F), 256>>>(dp, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("stagger %8lld B: %.1f us per family -> %.2f TB/s\n", (long long)stagger, ms / 5 / F * 1e3, (double)F * n * 20 / (ms / 5 * 1e-3) / 1e12);
        for (void *b : bufs) hipFree(b);
        hipFree(dp);
    }
    return 0;
}
