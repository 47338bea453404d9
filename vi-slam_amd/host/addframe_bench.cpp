// addframe_bench.cpp -- latency of the path the reference's GPU main drives: one VISystemGPU::AddFrameGPU call per camera frame
// (/root/reference/src/main_vi_slamGPU.cpp:118-123 -> src/VISystemGPU.cpp:137-175: Camera::Update -> addGPUKeyframe (detect, match,
// gradients, patch points) -> EstimatePoseFeatures -> Track) on the class surface of vislam_host.hpp, frames handed over as cv::Mat in
// pageable host memory like the reference's DataReader does.  Prints one JSON line: per-frame wall-clock percentiles, kernel launches and
// host <-> device waits per frame (vis_debug_counters).  bench.py runs it for the `single_frame_api` leg.
//   usage: addframe_bench <calibration.xml> <frames> [warmup]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <vector>
#include "VISystemGPU.hpp"

using namespace cv;
using namespace std;
using namespace vi;

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: addframe_bench <calibration.xml> <frames> [warmup]\n"); return 2; }
    const int n = atoi(argv[2]), warm = argc > 3 ? atoi(argv[3]) : 5;
    const int W = 752, H = 480, DIM = 2048; const unsigned long long SEED = 0xE0C00001ULL;
    if (cuda::getCudaEnabledDeviceCount() <= 0) { std::fprintf(stderr, "no device\n"); return 1; }
    cuda::setDevice(0);
    std::vector<uint8_t> canvas((size_t)DIM * DIM);
    vis_synth_canvas(canvas.data(), DIM, SEED);
    auto frame = [&](int t) { Mat m(H, W, CV_8U); vis_synth_frame(canvas.data(), DIM, SEED, t, W, H, m.data, W); return m; };
    std::vector<Mat> frames;
    for (int t = 0; t <= n + warm; t++) frames.push_back(frame(t));               // generated up front: the loop below times AddFrameGPU only
    std::streambuf* quiet = std::cout.rdbuf(); std::ostringstream sink; std::cout.rdbuf(sink.rdbuf());   // the adapters print the reference's messages
    VISystemGPU visystem(argc, argv);
    visystem.InitializeSystemGPU(argv[1], Point3d(0.1, -0.2, 0.3), Point3d(0.01, 0.02, -0.01), Point3d(0.02, -0.01, 0.3), frames[0]);
    vector<Point3d> gyro(10, Point3d(0, 0, 0)), acc(10, Point3d(0, 0, 9.81));
    for (int t = 1; t <= warm; t++) visystem.AddFrameGPU(frames[t], gyro, acc);
    unsigned long long c0[4], c1[4];
    vis_debug_counters(VisDevice::get(), c0);
    // per call: wall clock + what the call did (kernel launches, host waits, copies: vis_debug_counters deltas) -- the worst frame is the
    // figure that matters for a 20 Hz camera, and the deltas tell a slow wait from extra work (a keyframe insertion, a buffer growing)
    struct Call { double ms; int index; unsigned long long launches, waits, copies; int keyframes; };
    std::vector<Call> calls;
    unsigned long long ca[4], cb[4];
    for (int t = warm + 1; t <= warm + n; t++) {
        vis_debug_counters(VisDevice::get(), ca);
        const auto a = std::chrono::steady_clock::now();
        visystem.AddFrameGPU(frames[t], gyro, acc);
        const double d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
        vis_debug_counters(VisDevice::get(), cb);
        calls.push_back({d, t - warm - 1, cb[0] - ca[0], cb[1] - ca[1], cb[2] - ca[2], (int)visystem.cameraGPU.frameList.size()});
    }
    vis_debug_counters(VisDevice::get(), c1);
    std::cout.rdbuf(quiet);
    std::vector<double> ms; for (auto& c : calls) ms.push_back(c.ms);
    std::sort(ms.begin(), ms.end());
    double mean = 0; for (double v : ms) mean += v; mean /= ms.size();
    auto pct = [&](double q) { return ms[std::min(ms.size() - 1, (size_t)(ms.size() * q))]; };
    std::vector<Call> slow = calls;
    std::sort(slow.begin(), slow.end(), [](const Call& x, const Call& y) { return x.ms > y.ms; });
    std::printf("{\"frames\": %d, \"ms_p50\": %.4f, \"ms_p95\": %.4f, \"ms_p99\": %.4f, \"ms_max\": %.4f, \"slowest_frame_index\": %d, \"ms_mean\": %.4f, \"ms_min\": %.4f, "
                "\"launches_per_frame\": %.1f, \"host_waits_per_frame\": %.2f, \"async_copies_per_frame\": %.1f, \"keyframes\": %d, \"slowest_calls\": [",
                n, pct(0.5), pct(0.95), pct(0.99), ms.back(), slow[0].index, mean, ms[0],
                (double)(c1[0] - c0[0]) / n, (double)(c1[1] - c0[1]) / n, (double)(c1[2] - c0[2]) / n, (int)visystem.cameraGPU.frameList.size());
    for (size_t i = 0; i < std::min<size_t>(5, slow.size()); i++)
        std::printf("%s{\"index\": %d, \"ms\": %.4f, \"launches\": %llu, \"host_waits\": %llu, \"copies\": %llu, \"keyframes\": %d}", i ? ", " : "",
                    slow[i].index, slow[i].ms, slow[i].launches, slow[i].waits, slow[i].copies, slow[i].keyframes);
    std::printf("]}\n");
    return 0;
}
