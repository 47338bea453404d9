// vislam_host.hpp -- C++ adapters that keep the reference's class surface for the hot path and call
// HIP through the C ABI (include/vislam_hip.h).  Names, member names, argument meaning and call order
// mirror the reference headers:
//   Frame, Camera      /root/reference/include/Camera.hpp:32-144   (hot-path members only)
//   Matcher            /root/reference/include/Matcher.hpp:25-69
//   MatcherGPU         /root/reference/include/MatcherGPU.hpp:14-28
//   CameraGPU          /root/reference/include/CameraGPU.hpp:15-34
//   vi::VISystemGPU    /root/reference/include/VISystemGPU.hpp:14-36 (+ EstimatePoseFeaturesRansac of VISystem)
// Gradients and the patch point lists are included; IMU, ROS, Sophus poses, GUI are not.
#ifndef VISLAM_HOST_HPP_
#define VISLAM_HOST_HPP_
#include <iostream>
#include <string>
#include <vector>
#include "compat/opencv2/cudafeatures2d.hpp"
#include "../../include/vislam_hip.h"

using namespace cv;
using namespace std;

enum detectorType { USE_KAZE, USE_AKAZE, USE_ORB, USE_SIFT, USE_SURF };                       // include/Camera.hpp:21-28
enum matcherType { USE_BRUTE_FORCE, USE_BRUTE_FORCE_HAMMING, USE_FLANN, USE_BRUTE_FORCE_GPU, USE_BRUTE_FORCE_GPU_HAMMING };  // Matcher.hpp:15-23

// one device context shared by CameraGPU and its MatcherGPU (one per process/GPU, like cuda::setDevice)
struct VisDevice {
    static vis_ctx* get();
    static void fail(int rc, const char* where);        // reference style: cout + exit (src/CameraModel.cpp:46-52)
};

class Frame {                                            // include/Camera.hpp:32-68
public:
    Frame(); ~Frame();
    vector<Mat> grayImage = vector<Mat>(5);
    vector<Mat> gradientX = vector<Mat>(5), gradientY = vector<Mat>(5), gradient = vector<Mat>(5);   // :47-49 (CV_16S, CV_16S, CV_8U)
    vector<Mat> candidatePoints = vector<Mat>(5), candidateDebugPoints = vector<Mat>(5);               // :57,59 (N x 4 CV_32F rows)
    vector<KeyPoint> keypoints, prevGoodMatches, nextGoodMatches;
    Mat descriptors;
    int idFrame = 0; double imageTime = 0;
    bool obtainedGradients, obtainedGoodMatches, isKeyFrame;
    int gpuSlot = -1;                                    // device slot holding keypoints + descriptors
};

class Matcher {                                          // include/Matcher.hpp:25-69
public:
    Matcher(); Matcher(int _matcher);
    void setKeypoints(vector<KeyPoint> _keypoints_1, vector<KeyPoint> _keypoints_2);
    void setDescriptors(Mat _descriptors_1, Mat _descriptors_2);
    void setMatcher(int _matcher);
    void setImageDimensions(int w, int h);
    void computeMatches();
    void computeBestMatches(int n_features);
    void getMatches(vector<KeyPoint>& _matched1, vector<KeyPoint>& _matched2);
    void getGoodMatches(vector<KeyPoint>& _matched1, vector<KeyPoint>& _matched2);
    void printStatistics();
    void clear();
    vector<vector<DMatch> > aux_matches1, aux_matches2;
    vector<DMatch> matches, sortedMatches, goodMatches;
    vector<KeyPoint> keypoints_1, keypoints_2;
    Mat descriptors_1, descriptors_2;
    int h_size = 0, w_size = 0, nSymMatches = 0, nBestMatches = 0;
    double elapsed_knn1 = 0, elapsed_knn2 = 0, elapsed_symMatches = 0, elapsed_sortMatches = 0, elapsed_bestMatches = 0;
protected:
    int slot1 = -1, slot2 = -1;                          // set by CameraGPU when both descriptor sets are device resident
    friend class CameraGPU;
};

class MatcherGPU : public Matcher {                      // include/MatcherGPU.hpp:14-28
public:
    MatcherGPU(); MatcherGPU(int _matcher);
    void computeGPUMatches();
    void setGPUMatcher(int _matcher);
    bool useGPU = false;
    int matcherType = 0;
    cuda::GpuMat descriptorsGPU[2];
};

class Camera {                                           // include/Camera.hpp:70-144 (hot-path members)
public:
    Camera();
    void Update(Mat _grayImage);
    void computeGradient();                              // src/Camera.cpp:167-184
    void ObtainPatchesPointsPreviousFrame();             // src/Camera.cpp:358-410
    void ObtainDebugPointsPreviousFrame();               // src/Camera.cpp:413-445
    void saveFrame();
    void printStatistics();
    vector<Frame*> frameList;
    vector<DMatch> goodMatches;
    Frame* currentFrame = nullptr;
    int detectorType = 0, matcherType = 0, nPointsDetect = 0, nBestMatches = 0, n_cells = 0;
    vector<int> w_size = vector<int>(5), h_size = vector<int>(5);
    int w_patch = 0, h_patch = 0;
    double elapsed_detect = 0, elapsed_descriptors = 0, elapsed_computeGoodMatches = 0, elapsed_computeGradient = 0, elapsed_computePatches = 0;
    double elapsed_detect_mean = 0, elapsed_computeGoodMatches_mean = 0, nPointsDetect_mean = 0, nBestMatches_mean = 0;
    double elapsed_detect_sum = 0, elapsed_computeGoodMatches_sum = 0, nPointsDetect_sum = 0, nBestMatches_sum = 0;
    int num_images = 0;
};

class CameraGPU : public Camera {                        // include/CameraGPU.hpp:15-34
public:
    CameraGPU();
    CameraGPU(int _detector, int _matcher, int _w_size, int _h_size, int _num_cells, int _length_patch);
    void initializateCameraGPU(int _detector, int _matcher, int _w_size, int _h_size, int _num_cells, int _length_patch);
    void setGPUDetector(int _detector);
    void setGPUMatcher(int _matcher);
    int detectAndComputeGPUFeatures();
    void computeGPUGoodMatches();
    bool addGPUKeyframe();
    MatcherGPU matcherGPU;
    cuda::GpuMat frameGPU, keypointsGPU, descriptorsGPU;
    bool useGPU = false;
private:
    int nextSlot = 0;
};

class ImageReader {                                      // include/ImageReader.hpp:15-37 over vis_image_list / vis_image_read
public:
    ImageReader();
    ImageReader(string _directory);
    void setPath(string _directory);
    void setRawSize(int w, int h);                       // needed for headerless .raw files only
    string getImageName(int index);
    long int getImageTime(int index);
    void searchImages();
    Mat getImage(int index);                             // imread(.., CV_LOAD_IMAGE_GRAYSCALE) for P5 PGM / raw
    size_t getSize();
    void computeTimeStep();
    double TimeStep = 0.0;
private:
    string path;
    vector<string> file_names;
    int raw_w = 0, raw_h = 0;
};

namespace vi {
class VISystemGPU {                                      // include/VISystemGPU.hpp:14-36 (hot path only)
public:
    VISystemGPU();
    VISystemGPU(int argc, char* argv[]);
    ~VISystemGPU();
    // calibration: fx fy cx cy, image size, num_cells, detector, matcher (CameraModel fields, src/CameraModel.cpp:25-42)
    void InitializeSystemGPU(double fx, double fy, double cx, double cy, int w, int h, int num_cells, int detector, int matcher, Mat image);
    void InitializeCameraGPU(int _detector, int _matcher, int _w_size, int _h_size, int _num_cells, int _length_path);
    void AddFrameGPU(Mat _currentImage);                 // IMU arguments dropped (out of scope)
    void FreeLastFrameGPU();
    // VISystem::EstimatePoseFeaturesRansac, src/VISystem.cpp:1655-1708: R (3x3 row-major, f32), t (unit, f32)
    int EstimatePoseFeaturesRansac(Frame* _previous_frame, Frame* _current_frame, float R_out[9], float t_out[3]);
    CameraGPU cameraGPU;
    int num_keyframes = 0, num_max_keyframes = 20, min_features = 20;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    bool initialized = false;
    Mat currentImage, prevImage;
    int lastInliers = 0, lastPoseGood = 0;
};
}  // namespace vi
#endif
