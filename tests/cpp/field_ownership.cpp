// tests/cpp/field_ownership.cpp -- who owns the public fields of the calculator (include/opticalFlowCalc.h).
// The reference's settings thread writes m_deltaScalar & co. while the streaming thread is inside a blocking
// calculator call (HopperRender.cpp:1385-1390; the two threads share no lock) and NewSegment zeroes m_frameCount
// (:840): such a write must survive the call in flight and be used by the next one.
//   field_ownership <H> <W> <in_prefix> <flow_out>
// reads <in_prefix>{0,1,2}.bin (NV12), prints the fields, writes the blurred flow of a final calculateOpticalFlow().
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "opticalFlowCalcSDR.h"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int H = atoi(argv[1]), W = atoi(argv[2]);
    const std::string in = argv[3], flowOut = argv[4];
    const size_t bytes = (size_t)H * W + (size_t)(H / 2) * W;
    std::vector<std::vector<unsigned char>> frames(3, std::vector<unsigned char>(bytes));
    for (int k = 0; k < 3; k++) {
        std::ifstream f(in + std::to_string(k) + ".bin", std::ios::binary);
        f.read((char*)frames[k].data(), (std::streamsize)bytes);
        if (!f) return 3;
    }
    try {
        OpticalFlowCalcSDR calc(H, W, W, W, DEFAULT_DELTA_SCALAR, DEFAULT_NEIGHBOR_SCALAR, DEFAULT_BLACK_LEVEL, DEFAULT_WHITE_LEVEL, MAX_CALC_RES);
        for (int k = 0; k < 3; k++) calc.updateFrame(frames[k].data());
        std::atomic<bool> written{false};
        std::thread settings([&] {   // one write, somewhere inside the streaming thread's run of blocking calls
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            calc.m_deltaScalar = 5;
            calc.m_opticalFlowSearchRadius = 9;
            calc.m_outputWhiteLevel = 200.0f;
            written.store(true);
        });
        int calls = 0, after = 0;
        while (after < 200) {          // the streaming thread: blocking call after blocking call
            calc.calculateOpticalFlow();
            calls++;
            if (written.load()) after++;
        }
        settings.join();
        printf("calls %d\n", calls);
        printf("m_deltaScalar %d\n", calc.m_deltaScalar);
        printf("m_opticalFlowSearchRadius %d\n", calc.m_opticalFlowSearchRadius);
        printf("m_outputWhiteLevel %.1f\n", calc.m_outputWhiteLevel);
        calc.calculateOpticalFlow();   // must run with delta scalar 5, radius 9
        hf_stats st{};
        hf_get_stats(calc.context(), &st);
        std::vector<int16_t> flow(2 * (size_t)st.low_width * st.low_height);
        if (hf_read_blurred_flow(calc.context(), 1, flow.data()) != HF_OK) return 4;
        std::ofstream o(flowOut, std::ios::binary);
        o.write((const char*)flow.data(), (std::streamsize)(flow.size() * sizeof(int16_t)));
        printf("m_totalFrameDelta %u\n", calc.m_totalFrameDelta);
        // NewSegment (HopperRender.cpp:840) then the next source frame: the counter restarts at 1
        printf("m_frameCount_before %u\n", calc.m_frameCount);
        calc.m_frameCount = 0;
        calc.updateFrame(frames[0].data());
        printf("m_frameCount_after_new_segment %u\n", calc.m_frameCount);
    } catch (const std::exception& e) {
        fprintf(stderr, "exception: %s\n", e.what());
        return 5;
    }
    return 0;
}
