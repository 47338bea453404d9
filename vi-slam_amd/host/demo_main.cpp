// demo_main.cpp -- minimal stand-in for the frame loop of /root/reference/src/main_vi_slamGPU.cpp:41-123
// (device query, VISystemGPU construction/initialisation, while-loop of AddFrameGPU) on the synthetic
// stream; prints one line per frame so tests can compare it with the oracle.
#include <cstdio>
#include "vislam_host.hpp"

int main(int argc, char** argv) {
    int nframes = argc > 1 ? atoi(argv[1]) : 4;
    int n_cuda_devices = cuda::getCudaEnabledDeviceCount();                       // main_vi_slamGPU.cpp:41
    if (n_cuda_devices > 0) cuda::setDevice(0);
    else { cout << "No CUDA device detected" << endl << "Exiting..." << endl; return -1; }
    const int W = 752, H = 480, DIM = 2048;
    std::vector<uint8_t> canvas((size_t)DIM * DIM);
    vis_synth_canvas(canvas.data(), DIM, 0xE0C00001ULL);
    Mat image(H, W, CV_8U);
    vis_synth_frame(canvas.data(), DIM, 0xE0C00001ULL, 0, W, H, image.data, W);
    vi::VISystemGPU visystem(argc, argv);                                         // :64
    visystem.InitializeSystemGPU(458.654, 457.296, 367.215, 248.375, W, H, 49, USE_ORB, USE_BRUTE_FORCE_GPU_HAMMING, image);   // :65
    for (int j = 0; j < nframes; j++) {                                           // :118-123
        Mat frame(H, W, CV_8U);
        vis_synth_frame(canvas.data(), DIM, 0xE0C00001ULL, j, W, H, frame.data, W);
        visystem.AddFrameGPU(frame);
        Frame* f = visystem.cameraGPU.frameList.back();
        std::printf("FRAME %d kps %d sym %d good %d inliers %d posegood %d\n", j, (int)f->keypoints.size(),
                    visystem.cameraGPU.matcherGPU.nSymMatches, (int)f->prevGoodMatches.size(), visystem.lastInliers, visystem.lastPoseGood);
        // the step after matching (computeGradient + patch builders): checksums of what addGPUKeyframe stored
        unsigned long long gsum = 0; long long gxsum = 0;
        for (int l = 0; l < 5; l++) {
            const Mat& g = f->gradient[l]; const Mat& gx = f->gradientX[l];
            for (int y = 0; y < g.rows; y++) for (int x = 0; x < g.cols; x++) { gsum += g.at<uint8_t>(y, x); gxsum += gx.at<int16_t>(y, x) * (long long)(1 + ((x + y) & 3)); }
        }
        int npatch = 0, ndebug = 0;
        if (visystem.cameraGPU.frameList.size() > 1) {
            Frame* prev = visystem.cameraGPU.frameList[visystem.cameraGPU.frameList.size() - 2];
            for (int l = 0; l < 5; l++) { npatch += prev->candidatePoints[l].rows; ndebug += prev->candidateDebugPoints[l].rows; }
        }
        std::printf("GRAD %d g %llu gx %lld patch %d debug %d\n", j, gsum, gxsum, npatch, ndebug);
    }
    return 0;
}
