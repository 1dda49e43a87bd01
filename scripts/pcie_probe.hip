// pcie_probe.hip -- measurement aid (not product): what the host link of the GPU box gives a staged pass.
//   * hipMemcpyAsync H2D from pinned memory, one and two streams, several chunk sizes
//   * a kernel that reads mapped pinned host memory directly: sequential 16-byte loads and 80-byte rows gathered
//     at scattered offsets (the shape of a device-driven fetch of packed SEQ rows)
// build: hipcc --offload-arch=gfx950 -O3 scripts/pcie_probe.hip -o build_variants/pcie_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_seq_read(const uint4 *src, size_t n16, unsigned long long *sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n16; i += stride) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x1234567ULL) *sink = acc;
}

// rows of `row16` 16-byte units; every `keep`-th row out of `of` is fetched, five lanes per row
__global__ void k_row_gather(const uint4 *src, size_t n_rows, int row16, int keep_mod, uint4 *dst, unsigned long long *sink) {
    const size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
    const size_t r = g / row16;
    const int c = (int)(g % row16);
    if (r >= n_rows) return;
    const size_t srow = r * (size_t)keep_mod + (r * 2654435761u) % keep_mod; // one row out of every keep_mod, jittered
    const uint4 v = src[srow * row16 + c];
    dst[r * row16 + c] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t GB = (size_t)1 << 30;
    const size_t total = (argc > 1 ? atoi(argv[1]) : 8) * GB;
    void *h = nullptr, *d = nullptr;
    double t0 = now();
    CK(hipHostMalloc(&h, total, hipHostMallocMapped));
    printf("hipHostMalloc %zu GiB: %.2f s\n", total / GB, now() - t0);
    t0 = now();
    memset(h, 1, total);
    printf("memset (first touch) %.2f s\n", now() - t0);
    CK(hipMalloc(&d, total));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    for (size_t chunk : {(size_t)4 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30}) {
        for (int ns = 1; ns <= 2; ns++) {
            CK(hipDeviceSynchronize());
            t0 = now();
            size_t off = 0;
            int k = 0;
            while (off < total) {
                const size_t n = off + chunk <= total ? chunk : total - off;
                CK(hipMemcpyAsync((char *)d + off, (char *)h + off, n, hipMemcpyHostToDevice, s[k % ns]));
                off += n;
                k++;
            }
            CK(hipDeviceSynchronize());
            const double dt = now() - t0;
            printf("H2D pinned chunk %4zu MiB streams %d: %.1f GB/s\n", chunk >> 20, ns, total / dt / 1e9);
        }
    }
    // D2H
    CK(hipDeviceSynchronize());
    t0 = now();
    CK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, s[0]));
    CK(hipDeviceSynchronize());
    printf("D2H pinned one call: %.1f GB/s\n", total / (now() - t0) / 1e9);
    // simultaneous H2D + D2H
    t0 = now();
    CK(hipMemcpyAsync(d, h, total / 2, hipMemcpyHostToDevice, s[0]));
    CK(hipMemcpyAsync((char *)h + total / 2, (char *)d + total / 2, total / 2, hipMemcpyDeviceToHost, s[1]));
    CK(hipDeviceSynchronize());
    printf("H2D + D2H together: %.1f GB/s total\n", total / (now() - t0) / 1e9);
    // pageable
    {
        const size_t pn = 2 * GB < total ? 2 * GB : total;
        void *p = malloc(pn);
        memset(p, 2, pn);
        t0 = now();
        CK(hipMemcpy(d, p, pn, hipMemcpyHostToDevice));
        printf("H2D pageable: %.1f GB/s\n", pn / (now() - t0) / 1e9);
        free(p);
    }
    // zero-copy kernels
    void *hd = nullptr;
    CK(hipHostGetDevicePointer(&hd, h, 0));
    unsigned long long *sink;
    CK(hipMalloc(&sink, 8));
    for (int grid : {1024, 4096, 16384}) {
        CK(hipDeviceSynchronize());
        t0 = now();
        hipLaunchKernelGGL(k_seq_read, dim3(grid), dim3(256), 0, s[0], (const uint4 *)hd, total / 16, sink);
        CK(hipDeviceSynchronize());
        printf("zero-copy sequential read grid %5d: %.1f GB/s\n", grid, total / (now() - t0) / 1e9);
    }
    for (int row16 : {5, 2, 1}) {
        for (int keep_mod : {1, 3}) {
            const size_t rows_total = total / 16 / row16;
            const size_t n_rows = rows_total / keep_mod;
            CK(hipDeviceSynchronize());
            t0 = now();
            const size_t lanes = n_rows * row16;
            hipLaunchKernelGGL(k_row_gather, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, s[0], (const uint4 *)hd, n_rows, row16,
                               keep_mod, (uint4 *)d, sink);
            CK(hipDeviceSynchronize());
            const double dt = now() - t0;
            printf("zero-copy gather rows of %3d B, 1 of %d: %.1f GB/s useful\n", row16 * 16, keep_mod, n_rows * row16 * 16.0 / dt / 1e9);
        }
    }
    return 0;
}
