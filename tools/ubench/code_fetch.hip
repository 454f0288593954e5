// tools/ubench/code_fetch.hip -- what the instructions of a kernel cost to FETCH: straight-line kernels of 4 ... 96 KB of 8-byte vector
// instructions (four independent chains, so a wave issues one per pass), one launch of 256 x WAVES waves, timed by the launch's own
// start / stop events: back to back with itself (code warm in the 64 KB instruction cache a pair of CUs shares, and in L2) against
// behind an "evictor" launch of 128 KB of different code (cold instruction cache, code in L2) and behind an L2 flush as well.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/code_fetch tools/ubench/code_fetch.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <utility>
#define I4 asm volatile("v_mad_u32_u24 %0, %0, 3, %0\n v_mad_u32_u24 %1, %1, 5, %1\n v_mad_u32_u24 %2, %2, 7, %2\n v_mad_u32_u24 %3, %3, 9, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#define I32 I4 I4 I4 I4 I4 I4 I4 I4
#define I128 I32 I32 I32 I32            // 128 instructions = 1 KB
#define K1 I128
#define K4 K1 K1 K1 K1
#define K16 K4 K4 K4 K4
template <int KB, int SALT>
__global__ __launch_bounds__(256) void code_k(unsigned* out, unsigned seed) {
    unsigned a = threadIdx.x + seed + SALT, b = a * 3u, c = a * 5u, d = a * 7u;
    if constexpr (KB >= 64) { K16 K16 K16 K16 }
    if constexpr (KB % 64 >= 32) { K16 K16 }
    if constexpr (KB % 32 >= 16) { K16 }
    if constexpr (KB % 16 >= 8) { K4 K4 }
    if constexpr (KB % 8 >= 4) { K4 }
    if constexpr (KB >= 128) { K16 K16 K16 K16 }
    if (a + b + c + d == 0x12345u) out[0] = a;
}
__global__ void flush_k(unsigned* p, size_t n) { for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] += 1; }

static hipEvent_t e0, e1;
template <int KB>
static float one(unsigned* out, int waves_per_wg, int wgs) {
    hipExtLaunchKernelGGL((code_k<KB, 0>), dim3(wgs), dim3(64 * waves_per_wg), 0, 0, e0, e1, 0, out, 1u);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
template <int KB>
static void run(unsigned* out, unsigned* big, size_t nbig, int waves_per_wg, int wgs) {
    std::vector<float> warm, cold, coldl2;
    one<KB>(out, waves_per_wg, wgs);
    for (int r = 0; r < 9; r++) warm.push_back(one<KB>(out, waves_per_wg, wgs));
    for (int r = 0; r < 9; r++) {
        code_k<128, 1><<<512, 256>>>(out, 2u);                       // 128 KB of other code through every instruction cache
        cold.push_back(one<KB>(out, waves_per_wg, wgs));
    }
    for (int r = 0; r < 5; r++) {
        code_k<128, 1><<<512, 256>>>(out, 2u);
        flush_k<<<4096, 256>>>(big, nbig);                           // 1 GB through the L2s and the Infinity Cache
        coldl2.push_back(one<KB>(out, waves_per_wg, wgs));
    }
    auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const float w = med(warm), c = med(cold), c2 = med(coldl2);
    printf("%3d KB  %d waves/wg x %4d wgs: warm %7.2f us  cold I$ %7.2f us (+%5.2f us = %5.3f us/KB)  cold I$ + L2 %7.2f us (+%5.2f us)\n", KB, waves_per_wg, wgs, w, c, c - w, (c - w) / KB, c2, c2 - w);
}
// Twelve launches back to back in one stream, as a refinement chain makes them: twelve DIFFERENT kernels of KB each (12 x KB > 64 KB: each finds
// the instruction cache cold, the code in L2) against twelve launches of ONE kernel.
template <int KB, int S> static void launch_one(unsigned* out, int wpw, int wgs) { code_k<KB, S + 10><<<wgs, 64 * wpw>>>(out, 1u); }
template <int KB, int... S>
static void chain_distinct(unsigned* out, int wpw, int wgs, std::integer_sequence<int, S...>) {
    (launch_one<KB, S>(out, wpw, wgs), ...);
}
template <int KB>
static void chain(unsigned* out, int wpw, int wgs) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best[2] = {1e9f, 1e9f};
    for (int r = 0; r < 12; r++) for (int kind = 0; kind < 2; kind++) {
        hipEventRecord(a);
        if (kind) chain_distinct<KB>(out, wpw, wgs, std::make_integer_sequence<int, 12>{});
        else for (int i = 0; i < 12; i++) launch_one<KB, 0>(out, wpw, wgs);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r >= 2) best[kind] = std::min(best[kind], ms * 1e3f);
    }
    printf("chain of 12 x %2d KB, %d waves/wg x %4d wgs: one kernel %7.2f us, twelve kernels %7.2f us: +%5.2f us per launch\n", KB, wpw, wgs, best[0], best[1], (best[1] - best[0]) / 12);
}
int main() {
    unsigned *out, *big; const size_t nbig = (size_t)1 << 28;
    hipMalloc(&out, 4096); hipMalloc(&big, nbig * 4); hipMemset(big, 0, nbig * 4);
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpw : {1, 4}) for (int wgs : {256, 2048}) {
        run<4>(out, big, nbig, wpw, wgs); run<8>(out, big, nbig, wpw, wgs); run<16>(out, big, nbig, wpw, wgs); run<32>(out, big, nbig, wpw, wgs);
        run<48>(out, big, nbig, wpw, wgs); run<64>(out, big, nbig, wpw, wgs); run<96>(out, big, nbig, wpw, wgs);
    }
    for (int wpw : {1, 4}) for (int wgs : {256, 1024}) { chain<4>(out, wpw, wgs); chain<8>(out, wpw, wgs); chain<16>(out, wpw, wgs); chain<32>(out, wpw, wgs); }
    return 0;
}
