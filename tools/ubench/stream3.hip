// tools/ubench/stream3.hip -- what a trivial, perfectly aligned 2-read + 1-write kernel reaches at the warp
// kernel's size (3 x 24.9 MB), i.e. the practical ceiling for warpFrames' traffic shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void blend3(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { uint4 x = a[i], y = b[i]; o[i] = make_uint4(x.x ^ y.x, x.y + y.y, x.z ^ y.z, x.w + y.w); }
}
__global__ __launch_bounds__(256) void blend3x2(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, size_t n) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < n) { uint4 x0 = a[i], x1 = a[i + 1], y0 = b[i], y1 = b[i + 1];
        o[i] = make_uint4(x0.x ^ y0.x, x0.y + y0.y, x0.z ^ y0.z, x0.w + y0.w); o[i + 1] = make_uint4(x1.x ^ y1.x, x1.y + y1.y, x1.z ^ y1.z, x1.w + y1.w); }
}
__global__ __launch_bounds__(256) void copy2(const uint4* __restrict__ a, uint4* __restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = a[i];
}
int main() {
    const size_t F = 24883200, n = F / 16;
    char *base; CK(hipMalloc(&base, 4 * F + (64 << 20)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t pad : {(size_t)0, (size_t)4096, (size_t)(1 << 20) + 8192}) {
        uint4* a = (uint4*)base; uint4* b = (uint4*)(base + F + pad); uint4* o = (uint4*)(base + 2 * (F + pad));
        for (int variant = 0; variant < 3; variant++) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CK(hipEventRecord(e0, s));
                for (int k = 0; k < 20; k++) {
                    if (variant == 0) blend3<<<(n + 255) / 256, 256, 0, s>>>(a, b, o, n);
                    else if (variant == 1) blend3x2<<<(n / 2 + 255) / 256, 256, 0, s>>>(a, b, o, n);
                    else copy2<<<(n + 255) / 256, 256, 0, s>>>(a, o, n);
                }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double us = best * 1e3 / 20, bytes = variant == 2 ? 2.0 * F : 3.0 * F;
            printf("pad %8zu %-9s %.2f us  %.0f GB/s\n", pad, variant == 0 ? "blend3" : variant == 1 ? "blend3x2" : "copy2", us, bytes / us / 1e3);
        }
    }
    return 0;
}
