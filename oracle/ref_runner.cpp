// oracle/ref_runner.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A script-driven command-line driver for the REFERENCE's own OpticalFlowCalc{SDR,HDR}
// classes (compiled, unmodified, from /root/reference/HopperRender into
// oracle/_ref/libhopperrender_ref.so by oracle/Makefile).  It replays the calls the
// DirectShow filter makes (reference HopperRender.cpp:907-1189) and dumps the reference's
// device buffers, so that tests can (a) pin oracle/hf_oracle.c to the reference and
// (b) produce the committed golden vectors under tests/golden/.  The reference needs a
// real OpenCL device: this binary only does useful work on the GPU box.
//
// Script grammar (one command per line, '#' comments):
//   create <hdr 0|1> <H> <W> <inStride> <outStride> <delta> <neighbor> <black> <white> <maxCalcRes>
//   radius <R>                   m_opticalFlowSearchRadius = R
//   params <delta> <neighbor> <black> <white>
//   framecount <n>               m_frameCount = n      (what NewSegment does, HopperRender.cpp:840)
//   update <file>                updateFrame(bytes of file)
//   calc                         calculateOpticalFlow()
//   warp <t> <mode>              warpFrames(t, mode)
//   copy                         copyFrame()
//   download <file>              downloadFrame -> file
//   dump_offsets <file>          raw int16 [2][lh][lw] of m_offsetArray
//   dump_blurred <idx> <file>    raw int16 [2][lh][lw] of m_blurredOffsetArray[idx]
//   stats                        prints one JSON line with the public fields
//   time_calc <n>                n x calculateOpticalFlow(), prints wall ms per call
//   time_warp <n> <t> <mode>     n x warpFrames(), clFinish, prints wall ms per call
//   destroy
// std headers first: the reference's opticalFlowCalc.h:14 defines a function-like `max` macro.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <sstream>
#include <vector>

#include "opticalFlowCalcSDR.h"
#include "opticalFlowCalcHDR.h"

static std::vector<unsigned char> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "ref_runner: cannot read %s\n", path.c_str()); exit(3); }
    return std::vector<unsigned char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& path, const void* p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)p, (std::streamsize)n);
    if (!f) { fprintf(stderr, "ref_runner: cannot write %s\n", path.c_str()); exit(3); }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_runner <script>\n"); return 2; }
    std::ifstream in(argv[1]);
    if (!in) { fprintf(stderr, "ref_runner: cannot open %s\n", argv[1]); return 2; }
    OpticalFlowCalc* c = nullptr;
    bool hdr = false;
    std::string line;
    try {
        while (std::getline(in, line)) {
            std::istringstream ss(line);
            std::string cmd;
            if (!(ss >> cmd) || cmd[0] == '#') continue;
            if (cmd == "create") {
                int h, H, W, is, os, d, n, m; float b, w;
                ss >> h >> H >> W >> is >> os >> d >> n >> b >> w >> m;
                hdr = h != 0;
                if (hdr) c = new OpticalFlowCalcHDR(H, W, is, os, d, n, b, w, m);
                else     c = new OpticalFlowCalcSDR(H, W, is, os, d, n, b, w, m);
                continue;
            }
            if (!c) { fprintf(stderr, "ref_runner: '%s' before create\n", cmd.c_str()); return 2; }
            const size_t bpp = hdr ? 2 : 1;
            const size_t N = (size_t)c->m_opticalFlowFrameWidth * c->m_opticalFlowFrameHeight;
            if (cmd == "radius") { ss >> c->m_opticalFlowSearchRadius; }
            else if (cmd == "params") { ss >> c->m_deltaScalar >> c->m_neighborBiasScalar >> c->m_outputBlackLevel >> c->m_outputWhiteLevel; }
            else if (cmd == "framecount") { ss >> c->m_frameCount; }
            else if (cmd == "update") {
                std::string f; ss >> f;
                std::vector<unsigned char> buf = slurp(f);
                const size_t need = bpp * ((size_t)c->m_frameHeight * c->m_inputStride + (size_t)(c->m_frameHeight / 2) * c->m_inputStride);
                if (buf.size() < need) { fprintf(stderr, "ref_runner: %s too small\n", f.c_str()); return 3; }
                c->updateFrame(buf.data());
            }
            else if (cmd == "calc") { c->calculateOpticalFlow(); }
            else if (cmd == "warp") { float t; int m; ss >> t >> m; c->warpFrames(t, m); }
            else if (cmd == "copy") { c->copyFrame(); }
            else if (cmd == "download") {
                std::string f; ss >> f;
                std::vector<unsigned char> buf(bpp * ((size_t)c->m_frameHeight * c->m_outputStride + (size_t)(c->m_frameHeight / 2) * c->m_outputStride));
                c->downloadFrame(buf.data());
                spit(f, buf.data(), buf.size());
            }
            else if (cmd == "dump_offsets" || cmd == "dump_blurred") {
                int idx = 0; std::string f;
                if (cmd == "dump_blurred") ss >> idx;
                ss >> f;
                std::vector<short> buf(2 * N);
                cl_mem m = cmd == "dump_offsets" ? c->m_offsetArray : c->m_blurredOffsetArray[idx];
                cl_int err = clEnqueueReadBuffer(c->m_queue, m, CL_TRUE, 0, buf.size() * sizeof(short), buf.data(), 0, NULL, NULL);
                if (err) { fprintf(stderr, "ref_runner: read failed %d\n", err); return 4; }
                spit(f, buf.data(), buf.size() * sizeof(short));
            }
            else if (cmd == "stats") {
                printf("{\"frame_count\": %u, \"total_frame_delta\": %u, \"search_radius\": %d, \"res_scalar\": %d, "
                       "\"low_w\": %d, \"low_h\": %d, \"ofc_calc_time\": %.9f, \"warp_calc_time\": %.9f}\n",
                       c->m_frameCount, c->m_totalFrameDelta, c->m_opticalFlowSearchRadius, c->m_opticalFlowResScalar,
                       c->m_opticalFlowFrameWidth, c->m_opticalFlowFrameHeight, c->m_ofcCalcTime, c->m_warpCalcTime);
            }
            else if (cmd == "time_calc") {
                int n; ss >> n;
                clFinish(c->m_queue);
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; i++) c->calculateOpticalFlow();
                clFinish(c->m_queue);
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
                printf("{\"time_calc_ms\": %.6f, \"n\": %d, \"search_radius\": %d}\n", ms, n, c->m_opticalFlowSearchRadius);
            }
            else if (cmd == "time_warp") {
                int n, m; float t; ss >> n >> t >> m;
                clFinish(c->m_queue);
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; i++) c->warpFrames(t, m);
                clFinish(c->m_queue);
                double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / n;
                printf("{\"time_warp_ms\": %.6f, \"n\": %d}\n", ms, n);
            }
            else if (cmd == "destroy") { delete c; c = nullptr; }
            else { fprintf(stderr, "ref_runner: unknown command '%s'\n", cmd.c_str()); return 2; }
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "ref_runner: reference threw: %s\n", e.what());
        return 5;
    }
    delete c;
    fflush(stdout);
    return 0;
}
