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

// same traffic with the warp kernel's work decomposition: wave = 2 rows x 1 KB of a 3840 x (2160 + 1080) x 2 B frame, 4 waves per
// workgroup along the row, optional XCD banding; nread source frames, 5 outputs; `dep` = dependent table lookups first
__global__ __launch_bounds__(256) void k_tile(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* o0, uint4* o1, uint4* o2, uint4* o3, uint4* o4,
                                              const unsigned* __restrict__ tab, int nread, int banded, int dep) {
    const int wpr = 8, rows2 = (2160 + 1080) / 2, n_tiles = wpr * rows2, n_blocks = n_tiles / 4, per_band = (n_blocks + 7) / 8;
    const int blk = banded ? (blockIdx.x & 7) * per_band + (blockIdx.x >> 3) : blockIdx.x;
    const int tile = blk * 4 + (threadIdx.x >> 6);
    if (blk >= n_blocks || tile >= n_tiles) return;
    const int rg = tile / wpr, tx = tile - rg * wpr;
    const int lane = threadIdx.x & 63;
    size_t row_v = 3840 * 2 / 16;                       // uint4 per row
    unsigned d = 0;
    if (dep) { d = tab[(rg * 8 + tx) & 0xFFFF]; d = tab[(d + lane) & 0xFFFF]; d &= 1; }   // two dependent lookups (always 0 or 1)
    uint4* outs[5] = {o0, o1, o2, o3, o4};
#pragma unroll
    for (int k = 0; k < 5; k++) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const size_t i = (size_t)(rg * 2 + r) * row_v + tx * 64 + lane;
            uint4 v = a[i + d];
            if (nread > 1) { uint4 x = b[i + d]; v.x += x.x; v.y ^= x.y; v.z += x.z; v.w += x.w; }
            v.x += k;
            outs[k][i] = v;
        }
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
    unsigned* tab; CK(hipMalloc(&tab, 65536 * 4)); CK(hipMemset(tab, 0, 65536 * 4));
    for (int nread : {1, 2}) for (int banded : {0, 1}) for (int dep : {0, 1}) {
        const int per = nread + 5, reps = NB / per;
        auto go = [&]() {
            for (int r = 0; r < reps; r++) {
                uint4** p = &buf[r * per];
                k_tile<<<((8 * 1620 / 4 + 7) / 8) * 8, 256>>>(p[0], p[nread - 1], p[nread], p[nread + 1], p[nread + 2], p[nread + 3], p[nread + 4], tab, nread, banded, dep);
            }
        };
        go(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("tile kernel %dR + 5W banded=%d dep=%d : %7.1f us per launch  %7.1f GB/s\n", nread, banded, dep, us, per * (double)F / us / 1e3);
    }
    return 0;
}
