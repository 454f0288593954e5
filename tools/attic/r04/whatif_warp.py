"""tools/attic/r04/whatif_warp.py -- STOPWATCH builds of the staged period warp (wrong results, never shipped, never in the product
source): each variant removes ONE ingredient of warp_wg_body from a scratch copy of hf_kernels.hip and builds it into
hopperrender_amd/lib/exp/<name>/ (git-ignored).  Timed with tools/ab_warp.sh / tools/ab_bench.sh on one box; the time a variant saves
bounds what optimising that ingredient can buy.

    python tools/attic/r04/whatif_warp.py [name ...]     (no names: all)
"""
import os, subprocess, sys, concurrent.futures as cf
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SRC = open(os.path.join(R, "hopperrender_amd/csrc/hf_kernels.hip")).read()

def sub(s, old, new, count=1):
    assert s.count(old) >= 1, old[:60]
    return s.replace(old, new) if count == 0 else s.replace(old, new, count)

V = {}
# phase C as a rolled loop over the outputs (CORRECT results): 1/5 of the code of the hot loop
V["loop"] = lambda s: sub(s, """#pragma unroll
    for (int j = 0; j < kMaxWarpOutputs; j++) {
        if (j < n) {
            Src S;
            const uint2 d = sh.tab[j][ci];""", """#pragma unroll 1
    for (int j = 0; j < n; j++) {
        {
            Src S;
            const uint2 d = sh.tab[j][ci];""")
# no blend arithmetic: both runs are read, the stored value is their XOR
V["noblend"] = lambda s: sub(s, """                    float2v bl = __builtin_elementwise_fma(fa, s21, fb * s12);          // :176-177 as compiled on gfx950""",
                             """                    if (true) { v[k * GROUP + i] = S.ra[r][k].v[i] ^ S.rb[r][k].v[i]; v[k * GROUP + i + 1] = S.ra[r][k].v[i + 1] ^ S.rb[r][k].v[i + 1]; continue; }
                    float2v bl = __builtin_elementwise_fma(fa, s21, fb * s12);""")
# source B's runs are not read from LDS (its window is still copied)
V["nob"] = lambda s: sub(s, """                for (int r = 0; r < ROWS; r++) S.rb[r][0] = lds_run(p + (unsigned)r * rowb_b, off, odd);""",
                         """                for (int r = 0; r < ROWS; r++) S.rb[r][0] = S.ra[r][0];   (void)p;""")
# no window copy at all (LDS holds garbage)
V["nocopy"] = lambda s: sub(sub(s, "    if (need_a) stage(A, cmin_a, ymin_a, C_a, R_a, lds);", "    if (need_a && n == 77) stage(A, cmin_a, ymin_a, C_a, R_a, lds);"),
                            "    if (need_b) stage(B, cmin_b, ymin_b, C_b, R_b, lds + CHUNKS * 16);", "    if (need_b && n == 77) stage(B, cmin_b, ymin_b, C_b, R_b, lds + CHUNKS * 16);")
# no output stores from the staged path
V["nostore"] = lambda s: sub(s, """            E* __restrict__ out = (E*)a.outv[j] + out_off;
            warp_finish<E, VEC, ROWS, MODE, CZ, 16>(S, a.s12v[j], a.s21v[j], out, So, ROWS, lv);
        }
    }
}""", """            E* __restrict__ out = (E*)a.outv[j] + out_off;
            warp_finish<E, VEC, ROWS, MODE, CZ, 16>(S, a.s12v[j], a.s21v[j], out, So, n == 77 ? ROWS : 0, lv);
        }
    }
}""")
# no flow lookups and no displacement arithmetic: zero displacement everywhere
V["noA"] = lambda s: sub(sub(s, "            const uint32_t f12 = a.flow_xy[(size_t)ly * lw + lx];", "            const uint32_t f12 = n == 77 ? a.flow_xy[(size_t)ly * lw + lx] : 0u;"),
                         "            const uint32_t f21 = a.flow_xy[(size_t)py * lw + px];", "            const uint32_t f21 = n == 77 ? a.flow_xy[(size_t)py * lw + px] : 0u;")
# LDS run reads at conflict-free addresses (wrong data): lane-contiguous dwords
V["noconf"] = lambda s: sub(s, """        const uint32_t* p = (const uint32_t*)p8;
        uint32_t w[NDW + 1];
#pragma unroll
        for (int k = 0; k <= NDW; k++) w[k] = p[k];""", """        const uint32_t* p = (const uint32_t*)(lds + ((size_t)(p8 - lds) & 0x3000)) + lane;
        uint32_t w[NDW + 1];
#pragma unroll
        for (int k = 0; k <= NDW; k++) w[k] = p[64 * k];""")

# ONE flow round trip: the second (dependent) lookup takes the value of the first
V["f21"] = lambda s: sub(s, "            const uint32_t f21 = a.flow_xy[(size_t)py * lw + px];", "            const uint32_t f21 = n == 77 ? a.flow_xy[(size_t)py * lw + px] : f12 ^ (uint32_t)(px + py);")
# both lookups made, displacements forced to zero (windows = the tile, no motion)
V["zero"] = lambda s: sub(sub(s, "            const int ox12 = (int)(int16_t)(f12 & 0xFFFFu), oy12 = (int)(int16_t)(f12 >> 16);\n            const int py = clampi(ly - (oy12 >> rs), 0, lh - 1), px",
                                 "            const int keep = n == 77 ? 1 : 0;\n            const int ox12 = keep * (int)(int16_t)(f12 & 0xFFFFu), oy12 = keep * (int)(int16_t)(f12 >> 16);\n            const int py = clampi(ly - ((int)(int16_t)(f12 >> 16) >> rs), 0, lh - 1), px"),
                          "            const int ox21 = (int)(int16_t)(f21 & 0xFFFFu), oy21 = (int)(int16_t)(f21 >> 16);", "            const int ox21 = keep * (int)(int16_t)(f21 & 0xFFFFu), oy21 = keep * (int)(int16_t)(f21 >> 16);")
# every output store goes to one 1 MB window of its frame (L2-resident): the store INSTRUCTIONS without the HBM writes
V["st1m"] = lambda s: sub(s, "    const size_t out_off = (size_t)CZ * H * So + (size_t)cy0 * So + cx0;\n#pragma unroll\n    for (int j = 0; j < kMaxWarpOutputs; j++) {",
                          "    const size_t out_off = ((size_t)CZ * H * So + (size_t)cy0 * So + cx0) & (size_t)0x7FFF8;\n#pragma unroll\n    for (int j = 0; j < kMaxWarpOutputs; j++) {")

# the blend is computed but (almost) no store executes: the predicate depends on the blended data (v2 kernel: buffer stores)
V["stpred"] = lambda s: sub(s, "                __builtin_amdgcn_raw_buffer_store_b128(*(const buf_v4*)v, rsrc_out, out_off, (unsigned)r * out_pitch, 2 /* nt: streaming */);",
                            "                if (((const uint32_t*)v)[0] == 0x12345679u && ((const uint32_t*)v)[3] == 0x9abcdef1u) __builtin_amdgcn_raw_buffer_store_b128(*(const buf_v4*)v, rsrc_out, out_off, (unsigned)r * out_pitch, 2);")
# the stores execute, the blend does not (XOR of the two runs), LDS reads kept -- with "noblend" this separates arithmetic from stores
# windows as real motion makes them, but phase C reads its runs with ZERO displacement (aligned, inside the window: the tile itself lies in
# the union of the runs' extents only if some output has no displacement -- t = 0 for source A; so: source A only, B reads A's window)
V["czero"] = lambda s: sub(sub(s, "                const uint32_t w = base + d.x, wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u, off = CZ ? (wb & ~1u) : wb;\n                const unsigned char* p = win_a +",
                                  "                const uint32_t w = base + (n == 77 ? d.x : sh.tab[0][ci].x), wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u, off = CZ ? (wb & ~1u) : wb;\n                const unsigned char* p = win_a +"),
                           "                const uint32_t w = base + d.y, wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u, off = CZ ? (wb & ~1u) : wb;\n                const unsigned char* p = win_b +",
                           "                const uint32_t w = base + (n == 77 ? d.y : sh.tab[0][ci].y), wb = w & 0xFFFFu, odd = CZ ? (wb & 1u) : 0u, off = CZ ? (wb & ~1u) : wb;\n                const unsigned char* p = win_b +")
def build(name):
    d = os.path.join(R, "hopperrender_amd/lib/exp", "w_" + name); os.makedirs(d, exist_ok=True)
    src = os.path.join(d, "hf_kernels.hip")
    open(src, "w").write(V[name](SRC))
    F = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function".split()
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + F + ["-I", os.path.join(R, "include"), "-I", os.path.join(R, "hopperrender_amd/csrc"), "-c", src, "-o", src + ".o"], stderr=subprocess.DEVNULL)
    L = os.path.join(R, "hopperrender_amd/lib")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libhopperflow.so"), src + ".o"] +
                          [os.path.join(L, f) for f in ("hf_flow.hip.o", "hf_context.hip.o", "hf_calc.hip.o", "hf_batch.hip.o", "hf_async_io.hip.o", "hf_filter.cpp.o", "hf_hostio.cpp.o")] + ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
    os.remove(src + ".o"); os.remove(src)
    return name

names = sys.argv[1:] or list(V)
with cf.ThreadPoolExecutor(4) as ex:
    for n in ex.map(build, names): print("built w_" + n, flush=True)
