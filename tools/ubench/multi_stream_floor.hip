// tools/ubench/multi_stream_floor.hip -- do chains of small dependent kernels on K streams overlap?
// Each stream runs M graphs of NK dependent latency-bound kernels; prints the aggregate time per chain vs K.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_chain(const int* __restrict__ in, int* __restrict__ out, int depth) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int v = i;
    for (int d = 0; d < depth; d++) v = in[v & 0xFFFF] + i;   // dependent loads
    out[i] = v;
}
int main() {
    const int NK = 14, M = 60, KMAX = 12;
    std::vector<hipStream_t> st(KMAX);
    std::vector<int*> a(KMAX), b(KMAX);
    for (int k = 0; k < KMAX; k++) {
        CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        CK(hipMalloc(&a[k], 8 << 20)); CK(hipMalloc(&b[k], 8 << 20)); CK(hipMemset(a[k], 0, 8 << 20));
    }
    for (int wgs : {1, 135, 540, 2040}) for (int depth : {1, 4}) for (int use_graph : {1, 0}) {
        std::vector<hipGraphExec_t> ge(KMAX);
        for (int k = 0; k < KMAX; k++) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[k], hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < NK; i++) k_chain<<<wgs, 256, 0, st[k]>>>(a[k], b[k], depth);
            CK(hipStreamEndCapture(st[k], &g));
            CK(hipGraphInstantiate(&ge[k], g, nullptr, nullptr, 0));
            CK(hipGraphLaunch(ge[k], st[k]));
            hipGraphDestroy(g);
        }
        CK(hipDeviceSynchronize());
        printf("wgs %4d depth %d %s:", wgs, depth, use_graph ? "graph" : "eager");
        for (int K : {1, 2, 4, 6, 8, 12}) {
            auto t0 = std::chrono::steady_clock::now();
            for (int m = 0; m < M; m++)
                for (int k = 0; k < K; k++) {
                    if (use_graph) CK(hipGraphLaunch(ge[k], st[k]));
                    else for (int i = 0; i < NK; i++) k_chain<<<wgs, 256, 0, st[k]>>>(a[k], b[k], depth);
                }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("  K=%d %.1f", K, us / (M * K));
        }
        printf("   (us per %d-kernel chain, aggregate)\n", NK);
        for (int k = 0; k < KMAX; k++) hipGraphExecDestroy(ge[k]);
    }
    return 0;
}
