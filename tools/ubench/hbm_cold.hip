// tools/ubench/hbm_cold.hip -- streaming bandwidth when the working set does NOT fit the 256 MB Infinity Cache:
// what can a kernel that reads 2 frames and writes 5 (the fused period warp, 174 MB) reach at best?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr size_t F = 24883200;            // one 2160p P010 frame
constexpr size_t NV = F / 16;

__global__ __launch_bounds__(256) void k_rw(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* o0, uint4* o1, uint4* o2, uint4* o3, uint4* o4,
                                            int nread, int nwrite, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v = {1, 2, 3, 4};
        if (nread > 0) { uint4 x = a[i]; v.x += x.x; v.y += x.y; v.z ^= x.z; v.w += x.w; }
        if (nread > 1) { uint4 x = b[i]; v.x += x.x; v.y ^= x.y; v.z += x.z; v.w += x.w; }
        if (nwrite > 0) o0[i] = v;
        if (nwrite > 1) { v.x++; o1[i] = v; }
        if (nwrite > 2) { v.x++; o2[i] = v; }
        if (nwrite > 3) { v.x++; o3[i] = v; }
        if (nwrite > 4) { v.x++; o4[i] = v; }
        if (nwrite == 0 && v.x == 0x12345) o0[i] = v;
    }
}

int main() {
    const int NB = 84;                     // 84 x 24.9 MB = 2.1 GB of rotating buffers
    std::vector<uint4*> buf(NB);
    for (auto& p : buf) { CK(hipMalloc(&p, F)); CK(hipMemset(p, 1, F)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 8192, 24300}) for (auto rw : {std::pair<int,int>{1, 0}, {2, 0}, {0, 1}, {0, 5}, {1, 1}, {2, 1}, {2, 5}}) {
        const int nr = rw.first, nw = rw.second, per = nr + nw;
        const int reps = NB / per;
        auto go = [&]() {
            for (int r = 0; r < reps; r++) {
                uint4** p = &buf[r * per];
                uint4* q[7]; for (int k = 0; k < 7; k++) q[k] = p[k % per];
                k_rw<<<grid, 256>>>(q[0], q[nr > 1 ? 1 : 0], q[nr], q[nr + (nw > 1)], q[nr + 2 * (nw > 2)], q[nr + 3 * (nw > 3)], q[nr + 4 * (nw > 4)], nr, nw, NV);
            }
        };
        go(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("grid %5d  %dR + %dW : %7.1f us per launch  %7.1f GB/s\n", grid, nr, nw, us, per * (double)F / us / 1e3);
    }
    return 0;
}
