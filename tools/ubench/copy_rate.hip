// tools/ubench/copy_rate.hip -- what a plain streaming copy sustains on this device, by cache policy, loads in flight and grid size
// (build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/copy_rate tools/ubench/copy_rate.hip; run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4 __attribute__((ext_vector_type(4)));
template <int LD_NT, int ST_NT, int U>
__global__ __launch_bounds__(256) void copy_k(const v4* __restrict__ s, v4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++) r[k] = LD_NT ? __builtin_nontemporal_load(s + i + k * stride) : s[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) { if (ST_NT) __builtin_nontemporal_store(r[k], d + i + k * stride); else d[i + k * stride] = r[k]; }
    }
    for (; i < n; i += stride) d[i] = s[i];
}
template <int LD_NT, int ST_NT, int U>
static void run(const char* name, const v4* a, v4* b, size_t n, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; r++) {
        hipEventRecord(e0); copy_k<LD_NT, ST_NT, U><<<blocks, 256>>>(a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    printf("%-34s blocks %6d  %7.1f GB/s read + write\n", name, blocks, 2.0 * n * 16 / (best * 1e-3) / 1e9);
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    v4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0x5a, bytes); hipMemset(b, 0, bytes);
    for (int blocks : {2048, 8192, 32768, 131072}) {
        run<0, 0, 4>("plain loads, plain stores, 4 deep", a, b, n, blocks);
        run<0, 1, 4>("plain loads, nt stores, 4 deep", a, b, n, blocks);
        run<1, 1, 4>("nt loads, nt stores, 4 deep", a, b, n, blocks);
        run<1, 1, 8>("nt loads, nt stores, 8 deep", a, b, n, blocks);
        run<0, 1, 1>("plain loads, nt stores, 1 deep", a, b, n, blocks);
    }
    // read-only and write-only for the mix
    return 0;
}
