// tools/ubench/launch_floor.hip -- measures the dependent-kernel boundary cost on this box (eager vs hipGraph).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_empty(int* p) { if (threadIdx.x == 9999) p[0] = 1; }
__global__ void k_chain(const int* __restrict__ in, int* __restrict__ out, int depth) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int v = i;
    for (int d = 0; d < depth; d++) v = in[v & 0xFFFF] + i;   // dependent loads
    out[i] = v;
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *a, *b; CK(hipMalloc(&a, 8 << 20)); CK(hipMalloc(&b, 8 << 20)); CK(hipMemset(a, 0, 8 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 32;
    for (int wgs : {1, 510, 2040}) for (int depth : {-1, 1, 3}) {
        auto enqueue = [&]() { for (int i = 0; i < N; i++) { if (depth < 0) k_empty<<<wgs, 256, 0, s>>>(a); else k_chain<<<wgs, 256, 0, s>>>(a, b, depth); } };
        // eager
        enqueue(); CK(hipStreamSynchronize(s));
        float best = 1e9;
        for (int r = 0; r < 5; r++) { CK(hipEventRecord(e0, s)); enqueue(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
        // graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); enqueue(); CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        float bestg = 1e9;
        for (int r = 0; r < 5; r++) { CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < bestg) bestg = ms; }
        printf("wgs %4d depth %2d : eager %.2f us/kernel   graph %.2f us/kernel\n", wgs, depth, best * 1e3 / N, bestg * 1e3 / N);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
