// vislam_host.hpp -- C++ adapters that keep the reference's class surface for the hot path and call
// HIP through the C ABI (include/vislam_hip.h).  Names, member names, signatures, argument meaning and call
// order mirror the reference headers:
//   Frame, Camera      /root/reference/include/Camera.hpp:32-144   (hot-path members only)
//   Matcher            /root/reference/include/Matcher.hpp:25-69
//   MatcherGPU         /root/reference/include/MatcherGPU.hpp:14-28
//   CameraGPU          /root/reference/include/CameraGPU.hpp:15-34
//   vi::CameraModel    /root/reference/include/CameraModel.hpp:27-146 (calibration XML fields, no undistortion)
//   vi::VISystem       /root/reference/include/VISystem.hpp:33-155  (the members the GPU path and its main touch)
//   vi::VISystemGPU    /root/reference/include/VISystemGPU.hpp:14-36
//   Quaterniond + toQuaternion / toRPY / rotationMatrix2RPY / RPY2rotationMatrix   /root/reference/include/Plus.hpp:9-33
// so that the calls src/main_vi_slamGPU.cpp makes (:41-48, :64-65, :123-144) compile against this header unchanged
// (vi-slam_amd/host/main_calls_gpu.cpp holds them verbatim).  Out of scope and therefore absent: the IMU core (ROS topics,
// src/Imu.cpp), undistortion / ROI (calib3d), the CPU-only estimators, GUI calls.
#ifndef VISLAM_HOST_HPP_
#define VISLAM_HOST_HPP_
#include <iostream>
#include <string>
#include <vector>
#include "compat/opencv2/cudafeatures2d.hpp"
#include "../../include/vislam_hip.h"

using namespace cv;
using namespace std;

#define PYRAMID_LEVELS 5                                                                     // include/VISystem.hpp:27

enum detectorType { USE_KAZE, USE_AKAZE, USE_ORB, USE_SIFT, USE_SURF };                       // include/Camera.hpp:21-28
enum matcherType { USE_BRUTE_FORCE, USE_BRUTE_FORCE_HAMMING, USE_FLANN, USE_BRUTE_FORCE_GPU, USE_BRUTE_FORCE_GPU_HAMMING };  // Matcher.hpp:15-23

// ---- include/Plus.hpp (scalar helpers the GPU main calls: src/main_vi_slamGPU.cpp:67,126-127) --------------------
struct Quaterniond { double w, x, y, z; };
Quaterniond toQuaternion(double roll, double pitch, double yaw);
Point3d toRPY(const Quaterniond& q);
Point3d rotationMatrix2RPY(Matx33f rotationMatrix);
Matx33f RPY2rotationMatrix(Point3d rpy);
Mat RPYAndPosition2transformationMatrix(Point3d rpy, Point3d position);
Matx33f transformationMatrix2rotationMatrix(Mat transformationMatrix);                      // (the reference returns a Mat; every use converts it to Matx33f)
Point3d transformationMatrix2position(Mat transformationMatrix);

// one device context shared by CameraGPU and its MatcherGPU (one per process/GPU, like cuda::setDevice)
struct VisDevice {
    static vis_ctx* get();
    static void fail(int rc, const char* where);        // reference style: cout + exit (src/CameraModel.cpp:46-52)
};

namespace vi {
// stands in for Sophus::SE3f (typedef SE3, include/Options.hpp:53) on the library's vis_se3_* value operations
struct SE3 {
    struct Point { float v[3]; Point(float a = 0, float b = 0, float c = 0) { v[0] = a; v[1] = b; v[2] = c; }
                   float& x() { return v[0]; } float& y() { return v[1]; } float& z() { return v[2]; } float operator()(int i) const { return v[i]; } };
    struct Quaternion { float w_, x_, y_, z_; Quaternion(float w = 1, float x = 0, float y = 0, float z = 0) : w_(w), x_(x), y_(y), z_(z) {}
                        float w() const { return w_; } float x() const { return x_; } float y() const { return y_; } float z() const { return z_; } };
    vis_se3f v;
    SE3() { v.qx = v.qy = v.qz = 0; v.qw = 1; v.tx = v.ty = v.tz = 0; }
    SE3(const Quaternion& q, const Point& t) { v.qx = q.x(); v.qy = q.y(); v.qz = q.z(); v.qw = q.w(); v.tx = t(0); v.ty = t(1); v.tz = t(2); }
    SE3(const Matx33f& R, const Point& t) { vis_se3_from_rt(R.val, t.v, &v); }
    SE3 operator*(const SE3& o) const { SE3 r; vis_se3_mul(&v, &o.v, &r.v); return r; }
    static SE3 exp(const float a[6]) { SE3 r; vis_se3_exp(a, &r.v); return r; }
    Point translation() const { return Point(v.tx, v.ty, v.tz); }
    Quaternion unit_quaternion() const { return Quaternion(v.qw, v.qx, v.qy, v.qz); }
    Matx33f rotationMatrix() const { float M[16]; vis_se3_matrix(&v, M); Matx33f R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = M[4 * r + c]; return R; }
};
}  // namespace vi

class Frame {                                            // include/Camera.hpp:32-68
public:
    Frame(); ~Frame();
    vector<Mat> grayImage = vector<Mat>(5);
    vector<Mat> gradientX = vector<Mat>(5), gradientY = vector<Mat>(5), gradient = vector<Mat>(5);   // :47-49 (CV_16S, CV_16S, CV_8U)
    vector<Mat> candidatePoints = vector<Mat>(5), candidateDebugPoints = vector<Mat>(5);               // :57,59 (N x 4 CV_32F rows)
    vector<KeyPoint> keypoints, prevGoodMatches, nextGoodMatches;
    Mat descriptors;
    int idFrame = 0; double imageTime = 0;
    vi::SE3 rigid_transformation_;                       // :63
    bool obtainedGradients, obtainedGoodMatches, isKeyFrame;
    int gpuSlot = -1;                                    // device slot holding keypoints + descriptors ...
    unsigned gpuGen = 0;                                 // ... as long as the slot's generation still equals this
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
    void setGPUFrames(Mat _frame1, Mat _frame2);         // declared at :20, never defined in the reference: defined here (keeps the images)
    void computeGPUMatches();
    void setGPUMatcher(int _matcher);
    bool useGPU = false;
    int matcherType = 0;
    Ptr<cuda::DescriptorMatcher> matcherGPU;             // :23 -- handle of the device matcher (createBFMatcher(NORM_HAMMING))
    cuda::GpuMat descriptorsGPU[2];
    Mat frameGPU1, frameGPU2;
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
protected:
    // device keyframe slots: a frame occupies `nextSlot` while it is the current frame and keeps it only when saveFrame()
    // makes it a keyframe; frames that are dropped re-use the same slot, so the slot ring advances once per keyframe
    int nextSlot = 0;
    unsigned slotGen[32] = {0};
};

class CameraGPU : public Camera {                        // include/CameraGPU.hpp:15-34
public:
    CameraGPU();
    CameraGPU(int _detector, int _matcher, int _w_size, int _h_size, int _num_cells, int _length_patch);
    void initializateCameraGPU(int _detector, int _matcher, int _w_size, int _h_size, int _num_cells, int _length_patch);
    void setGPUDetector(int _detector);
    void detectGPUFeatures();                            // declared at :22, never defined in the reference: detect + describe, result kept on the device
    void setGPUMatcher(int _matcher);
    int detectAndComputeGPUFeatures();
    void computeGPUGoodMatches();
    bool addGPUKeyframe();
    MatcherGPU matcherGPU;
    cuda::GpuMat frameGPU, keypointsGPU, descriptorsGPU;
    bool useGPU = false;
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
class CameraModel {                                      // include/CameraModel.hpp:27-146, src/CameraModel.cpp:16-142
public:
    void GetCameraModel(string _calibrationPath);        // reads the cv::FileStorage XML fields of src/CameraModel.cpp:25-42
    const Mat& GetK() const { return output_intrinsic_camera_; }
    const Mat& GetOriginalK() const { return original_intrinsic_camera_; }
    const Mat& GetMap1() const { return map1_; }
    const Mat& GetMap2() const { return map2_; }
    int GetOutputWidth() const { return out_width_; }
    int GetOutputHeight() const { return out_height_; }
    int GetInputWidth() const { return in_width_; }
    int GetInputHeight() const { return in_height_; }
    bool IsValid() const { return valid_; }
    // what VISystem::CalculateROI (src/VISystem.cpp:162-205) looks for in the rectified first image, from the geometry of the
    // rectification map alone: first / last column of the middle row and first / last row of the middle column whose source
    // position falls inside the input image, +-5 px margin.  (x1, y1, x2, y2); see INTEGRATION.md for the remaining gap
    void RectifiedROI(int roi[4]) const;
    const float* DistCoeffs() const { return dist_coeffs_; }
    Mat imu2cam0Transformation;                          // 4x4 CV_32F
    float camera_frecuency = 0, imu_frecuency = 0;
    int min_features = 0, num_max_keyframes = 0, start_index = 0, use_gt = 0, use_ros = 0, num_cells = 0, length_patch = 0, detector = 0, matcher = 0;
private:
    Mat original_intrinsic_camera_ = Mat::zeros(3, 3, CV_32FC1), output_intrinsic_camera_ = Mat::zeros(3, 3, CV_32FC1), map1_, map2_;
    float input_calibration_[4] = {0, 0, 0, 0}, dist_coeffs_[4] = {0, 0, 0, 0};
    int in_width_ = 0, in_height_ = 0, out_width_ = 0, out_height_ = 0;
    bool valid_ = false;
};

class VISystem {                                         // include/VISystem.hpp:33-155 (what the GPU path and its main use)
public:
    VISystem();
    void Calibration(string _calibration_path);                                    // src/VISystem.cpp:208-221
    void InitializePyramid(int _width, int _height, Mat _K);                       // :1451-1493
    // Gauss-Newton using Forward Compositional Algorithm - Using features, :1113-1448 (on the device: vis_estimate_pose_features)
    void EstimatePoseFeatures(Frame* _previous_frame, Frame* _current_frame);
    // findEssentialMat + recoverPose, :1655-1708; returns the inlier count; R (3x3 row-major f32), t (unit, f32) in the members below
    int EstimatePoseFeaturesRansac(Frame* _previous_frame, Frame* _current_frame);
    void Track();                                                                  // :1567-1635
    void setGtRes(Mat TranslationResGT, Mat RotationGT);                            // :406 (ground-truth seed of the alignment)
    bool initialized = false, distortion_valid = false, depth_available = false;
    int num_keyframes = 0, num_max_keyframes = 0, min_features = 0, start_index = 0;
    Mat map1, map2;
    int h = 0, w = 0, h_input = 0, w_input = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    Point3d positionImu, velocityImu, accImu;
    Quaterniond qOrientationImu = {1, 0, 0, 0};
    Point3d RPYOrientationImu;
    Mat world2imuTransformation;
    Matx33f world2imuRotation;
    Point3d positionCam, velocityCam, accCam;
    Quaterniond qOrientationCam = {1, 0, 0, 0};
    Point3d RPYOrientationCam;
    Mat prev_world2camTransformation;
    Mat imu2camTransformation;
    Matx33f imu2camRotation;
    Point3d imu2camTranslation;
    SE3 final_poseCam, final_poseImu, current_poseCam, current_poseImu;
    CameraModel* camera_model = nullptr;
    Mat currentImage, prevImage, K;
    vector<int> w_ = vector<int>(PYRAMID_LEVELS), h_ = vector<int>(PYRAMID_LEVELS);
    vector<float> fx_ = vector<float>(PYRAMID_LEVELS), fy_ = vector<float>(PYRAMID_LEVELS), cx_ = vector<float>(PYRAMID_LEVELS), cy_ = vector<float>(PYRAMID_LEVELS);
    vector<float> invfx_ = vector<float>(PYRAMID_LEVELS), invfy_ = vector<float>(PYRAMID_LEVELS), invcx_ = vector<float>(PYRAMID_LEVELS), invcy_ = vector<float>(PYRAMID_LEVELS);
    vector<Mat> K_ = vector<Mat>(PYRAMID_LEVELS);
    Mat TranslationResidual = Mat::zeros(3, 1, CV_32FC1);
    Matx33f RotationResidual = Matx33f::eye(), RotationResCam = Matx33f::eye();
    Point3f translationResEst;
    // results of the last pose estimates
    vis_align_result lastAlignment;                                                // EstimatePoseFeatures
    float ransacR[9], ransacT[3]; int lastInliers = 0, lastPoseGood = 0;            // EstimatePoseFeaturesRansac
};

class VISystemGPU : public VISystem {                    // include/VISystemGPU.hpp:14-36
public:
    VISystemGPU();
    VISystemGPU(int argc, char* argv[]);
    ~VISystemGPU();
    void InitializeSystemGPU(string _calPath, Point3d _iniPosition, Point3d _iniVelocity, Point3d _iniRPY, Mat image);
    void InitializeCameraGPU(int _detector, int _matcher, int _w_size, int _h_size, int _num_cells, int _length_path);
    void AddFrameGPU(Mat _currentImage, vector<Point3d> _imuAngularVelocity, vector<Point3d> _imuAcceleration);
    void FreeLastFrameGPU();
    CameraGPU cameraGPU;
};
}  // namespace vi
#endif
