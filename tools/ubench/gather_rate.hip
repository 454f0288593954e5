// tools/ubench/gather_rate.hip -- L1-resident gather throughput by element size / alignment / address pattern.
// Question: what does an UNALIGNED u32/u64 per-lane load (the flow kernels' strip loads) cost next to aligned ones?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T> __device__ __forceinline__ unsigned long long sum(T v) { return (unsigned long long)v; }
template <> __device__ __forceinline__ unsigned long long sum<uint4>(uint4 v) { return v.x + v.y + v.z + v.w; }
template <typename T> __device__ __forceinline__ T ld(const unsigned char* p) { T v; __builtin_memcpy(&v, p, sizeof(T)); return v; }

// every lane issues ITER x 16 independent loads at base + lane * lane_stride + k * k_stride + shift  (all inside a 32 KB window)
template <typename T>
__global__ __launch_bounds__(256) void k_gather(const unsigned char* __restrict__ buf, unsigned long long* out, int lane_stride, int k_stride, int shift, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = buf + (size_t)blockIdx.x % 8 * 65536 + wave * 128 + shift;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
        T v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int off = (lane * lane_stride + (k + it) * k_stride) & 0x7FFF;
            v[k] = ld<T>(base + off);
        }
#pragma unroll
        for (int k = 0; k < 16; k++) acc += sum(v[k]);
    }
    if (acc == 0x1234567ull) out[0] = acc;
}

template <typename T>
int run(const char* name, const unsigned char* buf, unsigned long long* out, int lane_stride, int k_stride, int shift) {
    const int iters = 64, blocks = 256 * 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_gather<T><<<blocks, 256>>>(buf, out, lane_stride, k_stride, shift, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_gather<T><<<blocks, 256>>>(buf, out, lane_stride, k_stride, shift, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_loads = (double)blocks * 256 * iters * 16;
    const double clk = ms * 1e-3 * 2.4e9;   // nominal 2.4 GHz
    printf("%-44s %8.1f us  %6.2f lane-loads/clk/CU  (%5.1f wave-instr clk)\n", name, ms * 1e3, lane_loads / clk / 256, 64.0 / (lane_loads / clk / 256));
    return 0;
}

int main() {
    unsigned char* buf; unsigned long long* out;
    CK(hipMalloc(&buf, 8 * 65536 + 65536)); CK(hipMemset(buf, 1, 8 * 65536 + 65536)); CK(hipMalloc(&out, 64));
    // contiguous lanes (lane stride = element size), candidates 64 B apart
    run<unsigned char>("u8  contiguous lanes", buf, out, 1, 64, 0);
    run<unsigned short>("u16 contiguous lanes aligned", buf, out, 2, 64, 0);
    run<unsigned short>("u16 contiguous lanes +1", buf, out, 2, 64, 1);
    run<unsigned>("u32 contiguous lanes aligned", buf, out, 4, 64, 0);
    run<unsigned>("u32 contiguous lanes +1 byte", buf, out, 4, 64, 1);
    run<unsigned>("u32 contiguous lanes +2 bytes", buf, out, 4, 64, 2);
    run<unsigned long long>("u64 contiguous lanes aligned", buf, out, 8, 64, 0);
    run<unsigned long long>("u64 contiguous lanes +1 byte", buf, out, 8, 64, 1);
    run<unsigned long long>("u64 contiguous lanes +4 bytes", buf, out, 8, 64, 4);

    run<uint4>("u128 contiguous lanes aligned", buf, out, 16, 1024, 0);
    run<uint4>("u128 contiguous lanes +2 bytes", buf, out, 16, 1024, 2);
    run<uint4>("u128 contiguous lanes +4 bytes", buf, out, 16, 1024, 4);
    run<uint4>("u128 contiguous lanes +8 bytes", buf, out, 16, 1024, 8);
    // flow-kernel-like: 8 lanes per row (4 B apart), rows 640 B apart
    run<unsigned>("u32 8 lanes/row, rows 640B, aligned", buf, out, 4 + 0, 640, 0);
    // each lane its own row (stride 644: rows + 4 B)
    run<unsigned char>("u8  lane stride 641 (all different lines)", buf, out, 641, 64, 0);
    run<unsigned>("u32 lane stride 644 aligned (diff lines)", buf, out, 644, 64, 0);
    run<unsigned>("u32 lane stride 644 +1 byte", buf, out, 644, 64, 1);
    run<unsigned>("u32 lane stride 132 aligned", buf, out, 132, 64, 0);
    run<unsigned>("u32 lane stride 132 +1", buf, out, 132, 64, 1);
    return 0;
}
