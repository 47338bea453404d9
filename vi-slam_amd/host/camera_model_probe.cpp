// camera_model_probe.cpp -- prints what vi::CameraModel derives from a calibration XML (src/CameraModel.cpp:16-101) and the ROI
// VISystem would crop to (src/VISystem.cpp:162-205): tests/test_camera_model.py compares it with an independent numpy evaluation.
// Host-only: no device call.
#include <cstdio>
#include "vislam_host.hpp"
int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: camera_model_probe calibration.xml\n"); return 2; }
    vi::CameraModel m;
    m.GetCameraModel(argv[1]);
    const cv::Mat& K = m.GetK(); const cv::Mat& K0 = m.GetOriginalK();
    int roi[4] = {0, 0, m.GetOutputWidth() - 1, m.GetOutputHeight() - 1};
    if (m.IsValid()) m.RectifiedROI(roi);
    std::printf("{\"valid\": %d, \"K\": [%.9g, %.9g, %.9g, %.9g], \"K0\": [%.9g, %.9g, %.9g, %.9g], \"in\": [%d, %d], \"out\": [%d, %d], \"roi\": [%d, %d, %d, %d], "
                "\"dist\": [%.9g, %.9g, %.9g, %.9g], \"num_cells\": %d}\n",
                m.IsValid() ? 1 : 0, K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2),
                K0.at<float>(0, 0), K0.at<float>(1, 1), K0.at<float>(0, 2), K0.at<float>(1, 2), m.GetInputWidth(), m.GetInputHeight(),
                m.GetOutputWidth(), m.GetOutputHeight(), roi[0], roi[1], roi[2], roi[3],
                m.DistCoeffs()[0], m.DistCoeffs()[1], m.DistCoeffs()[2], m.DistCoeffs()[3], m.num_cells);
    return 0;
}
