// compat/DataReader.hpp -- stand-in for /root/reference/include/DataReader.hpp:9-55 on the SYNTHETIC stream (dataset I/O, src/DataReader.cpp /
// GroundTruth.cpp, is out of scope: DESIGN 6; the EuRoC image directory reader itself is built -- SURVEY 8(f) N3, csrc/ingest.hpp + ImageReader).
// Same constructor, UpdateDataReader and public members main_vi_slamGPU.cpp:58-63,121-127 touches.  image_path selects the stream:
// "" / "synthetic[:N]" = S-752, "parallax[:N]" = S-752P (two depth layers + moving objects); N = frames served (indexLastData = 210 + N,
// main starts at j = 210).  Ground truth is a constant pose, IMU samples are a body at rest.
#ifndef VISLAM_COMPAT_DATAREADER_HPP_
#define VISLAM_COMPAT_DATAREADER_HPP_
#include <cstdlib>
#include <string>
#include <vector>
#include "../vislam_host.hpp"
using namespace std;
using namespace cv;
class DataReader {
public:
    DataReader(string image_path, string /*imu_path*/, string /*gt_path*/, char /*separator*/) : canvas((size_t)DIM * DIM) {
        const size_t colon = image_path.find(':');
        parallax = image_path.compare(0, 8, "parallax") == 0;
        vis_synth_canvas(canvas.data(), DIM, SEED);
        gtPosition.push_back(Point3d(0.1, -0.2, 0.3)); gtLinearVelocity.push_back(Point3d(0.01, 0.02, -0.01)); gtRPY.push_back(Point3d(0.02, -0.01, 0.3));
        gtQuaternion.push_back(toQuaternion(0.02, -0.01, 0.3));
        imuAngularVelocity.assign(10, Point3d(0, 0, 0)); imuAcceleration.assign(10, Point3d(0, 0, 9.81));
        indexLastData = 210 + (colon == string::npos ? 0 : atoi(image_path.c_str() + colon + 1));
    }
    void UpdateDataReader(int index, int index2) {
        image1 = frame(index - 209); image2 = frame(index2 - 209);                     // main starts at j = 210 (:62): stream frame t = j - 209
        currentTimeMs = 50.0 * (index2 - 210);
    }
    vector<Point3d> imuAngularVelocity, imuAcceleration, gtPosition, gtLinearVelocity, gtRPY;
    vector<Quaterniond> gtQuaternion;
    Mat image1, image2;
    double currentTimeMs = 0;
    int indexLastData;
private:
    Mat frame(int t) {
        Mat m(H, W, CV_8U);
        if (parallax) vis_synth_frame_parallax(canvas.data(), DIM, SEED, t, W, H, m.data, W);
        else vis_synth_frame(canvas.data(), DIM, SEED, t, W, H, m.data, W);
        return m;
    }
    static const int W = 752, H = 480, DIM = 2048;
    static constexpr unsigned long long SEED = 0xE0C00001ULL;
    std::vector<uint8_t> canvas;
    bool parallax = false;
};
#endif
