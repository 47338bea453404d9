// main_calls_gpu.cpp -- the calls /root/reference/src/main_vi_slamGPU.cpp makes into the GPU class surface, VERBATIM, compiled
// against the adapters (vislam_host.hpp) to show that the reference main drops in unchanged at its call sites:
//   :41-48   device query / selection          (cuda::getCudaEnabledDeviceCount, cuda::setDevice)
//   :64-65   VISystemGPU construction + InitializeSystemGPU(calibrationFile, gtPosition[0], gtLinearVelocity[0], gtRPY[0], image1)
//   :66-67   "Initializate System", toQuaternion(...)
//   :118-123 the frame loop: UpdateDataReader, AddFrameGPU(image2, imuAngularVelocity, imuAcceleration)
//   :125-152 the public members main reads afterwards (imu2camTranslation, imu2camRotation, positionCam, qOrientationCam) and the CSV row
// Those blocks are copied character for character between the BEGIN/END markers below (tests/test_host_adapters_gpu.py
// checks them against the line numbers above when the reference tree is present).  What surrounds them is NOT the
// reference's: `DataReader` and `VisualizerMarker` are stand-ins for the two out-of-scope components main also uses
// (dataset I/O: src/DataReader.cpp; ROS markers: src/Visualizer.cpp; compat/DataReader.hpp, compat/Visualizer.hpp) -- the stand-in reader
// serves the synthetic stream and a constant ground truth -- and the argument parsing (cv::CommandLineParser, :26-39,50-54) is plain argv here.
// This file is the RUNNABLE twin (it prints per-frame records for the parity test); the reference's own file is compiled unchanged
// against the same compat tree by tests/test_reference_main_compiles.py.
#include <cstdio>
#include <fstream>
#include "DataReader.hpp"
#include "Visualizer.hpp"
#include "VISystemGPU.hpp"

using namespace cv;
using namespace std;
using namespace vi;
cuda::DeviceInfo device_info;                                                          // :23

// the stand-ins for the two out-of-scope components main also uses live under the reference's own header names (compat/DataReader.hpp:
// the synthetic stream + a constant ground truth; compat/Visualizer.hpp: VisualizerMarker without ROS), so that the reference's file
// itself compiles against them unchanged (tests/test_reference_main_compiles.py)

static void print_f32(const char* tag, const float* v, int n) {                        // exact float bits for the parity test
    std::printf("%s", tag);
    for (int i = 0; i < n; i++) { uint32_t u; std::memcpy(&u, &v[i], 4); std::printf(" %08x", u); }
    std::printf("\n");
}

int main( int argc, char** argv ){
    if (argc == 3 && string(argv[1]) == "--calibration-only") {                        // CPU-only check of the XML reader (no device needed)
        CameraModel cm; cm.GetCameraModel(argv[2]);
        const Mat& K = cm.GetK();
        std::printf("CAL in %d %d out %d %d K %.9g %.9g %.9g %.9g valid %d freq %.9g %.9g min_features %d num_max_keyframes %d start_index %d use_gt %d use_ros %d "
                    "num_cells %d length_patch %d detector %d matcher %d\n", cm.GetInputWidth(), cm.GetInputHeight(), cm.GetOutputWidth(), cm.GetOutputHeight(),
                    K.at<float>(0, 0), K.at<float>(1, 1), K.at<float>(0, 2), K.at<float>(1, 2), (int)cm.IsValid(), cm.camera_frecuency, cm.imu_frecuency,
                    cm.min_features, cm.num_max_keyframes, cm.start_index, cm.use_gt, cm.use_ros, cm.num_cells, cm.length_patch, cm.detector, cm.matcher);
        std::printf("IMU2CAM");
        for (int i = 0; i < 16; i++) std::printf(" %.9g", cm.imu2cam0Transformation.at<float>(i / 4, i % 4));
        std::printf("\n");
        return 0;
    }
    if (argc < 4) { cout << "usage: vislam_main_gpu <calibration.xml> <frames> <output.csv> [parallax]" << endl; return 2; }
    // BEGIN verbatim src/main_vi_slamGPU.cpp:40-48
    cout << "===================================================" << endl;
    int n_cuda_devices = cuda::getCudaEnabledDeviceCount();
    if (n_cuda_devices > 0) {
        cuda::setDevice(0);
    } else {
        cout << "No CUDA device detected" << endl;
        cout << "Exiting..." << endl;
        return -1;
    }
    // END verbatim

    string gtFile = "", imuFile = "";
    string imagesPath = argc > 4 ? argv[4] : "";
    string calibrationFile = argv[1];
    string outputFile = argv[3];
    char separator = ',';
    DataReader Data(imagesPath, imuFile, gtFile, separator);

    // BEGIN verbatim src/main_vi_slamGPU.cpp:62-67
    int j = 210;
    Data.UpdateDataReader(j-1, j);
    VISystemGPU visystem(argc, argv);
    visystem.InitializeSystemGPU( calibrationFile, Data.gtPosition[0], Data.gtLinearVelocity[0], Data.gtRPY[0], Data.image1);
    cout << "Initializate System"<<endl;
    Quaterniond qinit = toQuaternion(Data.gtRPY[0].x, Data.gtRPY[0].y, Data.gtRPY[0].z);
    // END verbatim
    (void)qinit;
    {
        const float ini7[7] = {visystem.final_poseCam.v.qx, visystem.final_poseCam.v.qy, visystem.final_poseCam.v.qz, visystem.final_poseCam.v.qw,
                               visystem.final_poseCam.v.tx, visystem.final_poseCam.v.ty, visystem.final_poseCam.v.tz};
        print_f32("INITPOSE", ini7, 7);
    }
    VisualizerMarker visualizer_gtCam("gtCam_poses", "/my_frame", 2000, ARROW, 0, Point3f(0.5, 0.5, 0.5),Point3f(0.0, 1.0, 0.0));      // :73
    VisualizerMarker visualizer_estCam("estCam_poses", "/my_frame", 2000, ARROW, 0, Point3f(0.5, 0.5, 0.5),Point3f(0.5, 0.5, 0.5));   // :75
    Quaterniond qOrientationCamGT;                                                     // :106-108
    Point3d RPYOrientationCamGT;
    Point3d positionCamGT;
    std::ofstream outputFilecsv;                                                       // :113
    Point3d zero;                                                                      // :115
    outputFilecsv.open(outputFile.c_str(), std::ofstream::out | std::ofstream::trunc);  // :117 (the reference hard-codes a home directory path)
    Data.indexLastData = 210 + atoi(argv[2]);

    // BEGIN verbatim src/main_vi_slamGPU.cpp:118-152
    while(j <Data.indexLastData)
    {  // Cambiar por constant
        Mat finalImage, finalImage2;
        Data.UpdateDataReader(j, j+1);
        j = j+1;
        visystem.AddFrameGPU(Data.image2, Data.imuAngularVelocity, Data.imuAcceleration);
       
        positionCamGT = Data.gtPosition.back()+visystem.imu2camTranslation;
        RPYOrientationCamGT =rotationMatrix2RPY(visystem.imu2camRotation*RPY2rotationMatrix(toRPY(Data.gtQuaternion.back()) ));
        qOrientationCamGT = toQuaternion(RPYOrientationCamGT.x, RPYOrientationCamGT.y, RPYOrientationCamGT.z);



         //visualizer_gtIMU.UpdateMessages(zero, Data.gtQuaternion.back());
         visualizer_gtCam.UpdateMessages(zero, qOrientationCamGT);
         //visualizer_estIMU.UpdateMessages(zero, visystem.qOrientationImu);
         visualizer_estCam.UpdateMessages(zero, visystem.qOrientationCam);
         
         cout<< " Current time = "<< Data.currentTimeMs <<" ms " <<endl;
        
        outputFilecsv <<  visystem.positionCam.x<<","
        <<visystem.positionCam.y<<","
        <<visystem.positionCam.z<<","
        <<visystem.qOrientationCam.x <<","
        <<visystem.qOrientationCam.y <<","
        <<visystem.qOrientationCam.z <<","
        <<visystem.qOrientationCam.w <<","
        <<  positionCamGT.x <<","
        <<  positionCamGT.y <<","
        <<  positionCamGT.z <<","
        <<  qOrientationCamGT.x <<","        
        <<  qOrientationCamGT.y <<","
        <<  qOrientationCamGT.z <<","
        <<  qOrientationCamGT.w
        <<endl;
        // END verbatim

        // ---- not part of the reference main: one machine-readable record per frame for tests/test_host_adapters_gpu.py
        Frame* f = visystem.cameraGPU.frameList.back();
        std::printf("FRAME %d kps %d sym %d good %d keyframes %d\n", j - 211, (int)f->keypoints.size(), visystem.cameraGPU.matcherGPU.nSymMatches,
                    (int)f->prevGoodMatches.size(), (int)visystem.cameraGPU.frameList.size());
        const vis_align_result& a = visystem.lastAlignment;
        std::printf("ALIGN %d it %d %d %d %d res %d %d %d %d\n", j - 211, a.iterations[3], a.iterations[2], a.iterations[1], a.iterations[0],
                    a.n_residuals[3], a.n_residuals[2], a.n_residuals[1], a.n_residuals[0]);
        const float pose7[7] = {a.pose.qx, a.pose.qy, a.pose.qz, a.pose.qw, a.pose.tx, a.pose.ty, a.pose.tz};
        print_f32("ALIGNPOSE", pose7, 7);
        const float fin7[7] = {visystem.final_poseCam.v.qx, visystem.final_poseCam.v.qy, visystem.final_poseCam.v.qz, visystem.final_poseCam.v.qw,
                               visystem.final_poseCam.v.tx, visystem.final_poseCam.v.ty, visystem.final_poseCam.v.tz};
        print_f32("FINALPOSE", fin7, 7);
        if (visystem.cameraGPU.frameList.size() > 1) {
            Frame* prev = visystem.cameraGPU.frameList[visystem.cameraGPU.frameList.size() - 2];
            const int ninl = visystem.EstimatePoseFeaturesRansac(prev, f);              // the essential-matrix path (src/VISystem.cpp:1655), commented out at :263
            std::printf("RANSAC %d inliers %d posegood %d\n", j - 211, ninl, visystem.lastPoseGood);
        }
    }
    outputFilecsv.close();
    return 0;
}
