// vislam_host.cpp -- adapter bodies; every method cites the reference body it mirrors.
#include "vislam_host.hpp"
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <sstream>

namespace { int g_device = 0; vis_ctx* g_ctx = nullptr; }
namespace cv { namespace cuda {
void setDevice(int device) { g_device = device; }
int currentDevice() { return g_device; }
}}

vis_ctx* VisDevice::get() {
    if (!g_ctx) {
        int rc = vis_create(g_device, &g_ctx);
        if (rc) { cout << "No HIP device detected (" << vis_strerror(rc) << ")" << endl << "Exiting..." << endl; exit(1); }  // main_vi_slamGPU.cpp:44-48
    }
    return g_ctx;
}
void VisDevice::fail(int rc, const char* where) {
    cout << where << ": " << vis_strerror(rc) << " " << (g_ctx ? vis_last_error(g_ctx) : "") << endl;
    exit(1);
}

// ---------------------------------------------------------------- Plus (src/Plus.cpp): Euler / quaternion / matrix helpers
Quaterniond toQuaternion(double roll, double pitch, double yaw) {                // src/Plus.cpp:3-21
    const double cy = cos(yaw * 0.5), sy = sin(yaw * 0.5), cr = cos(roll * 0.5), sr = sin(roll * 0.5), cp = cos(pitch * 0.5), sp = sin(pitch * 0.5);
    Quaterniond q;
    q.w = cy * cr * cp + sy * sr * sp; q.x = cy * sr * cp - sy * cr * sp;
    q.y = cy * cr * sp + sy * sr * cp; q.z = sy * cr * cp - cy * sr * sp;
    return q;
}
Point3d toRPY(const Quaterniond& q) {                                             // src/Plus.cpp:23-54
    const double sinr_cosp = 2.0 * (q.w * q.x + q.y * q.z), cosr_cosp = 1.0 - 2.0 * (q.x * q.x + q.y * q.y);
    const double sinp = 2.0 * (q.w * q.y - q.z * q.x);
    const double siny_cosp = 2.0 * (q.w * q.z + q.x * q.y), cosy_cosp = 1.0 - 2.0 * (q.y * q.y + q.z * q.z);
    Point3d a;
    a.x = atan2(sinr_cosp, cosr_cosp);
    a.y = fabs(sinp) >= 1 ? copysign(M_PI / 2, sinp) : asin(sinp);                // 90 degrees when out of range
    a.z = atan2(siny_cosp, cosy_cosp);
    return a;
}
Point3d rotationMatrix2RPY(Matx33f R) {                                           // src/Plus.cpp:56-83
    const double r11 = R(0, 0), r21 = R(1, 0), r31 = R(2, 0), r32 = R(2, 1), r33 = R(2, 2);
    Point3d a;
    a.z = atan2(r21, r11); a.y = atan2(-r31, sqrt(r32 * r32 + r33 * r33)); a.x = atan2(r32, r33);
    return a;
}
static void rpy_rotation(const Point3d& rpy, double m[9]) {                       // the ZYX product both Plus.cpp builders write out
    const double c1 = cos(rpy.x), s1 = sin(rpy.x), c2 = cos(rpy.y), s2 = sin(rpy.y), c3 = cos(rpy.z), s3 = sin(rpy.z);
    m[0] = c3 * c2; m[1] = c3 * s2 * s1 - s3 * c1; m[2] = c3 * s2 * c1 + s3 * s1;
    m[3] = s3 * c2; m[4] = s3 * s2 * s1 + c3 * c1; m[5] = s3 * s2 * c1 - c3 * s1;
    m[6] = -s2;     m[7] = c2 * s1;                m[8] = c2 * c1;
}
Matx33f RPY2rotationMatrix(Point3d rpy) {                                         // src/Plus.cpp:182-220
    double m[9]; rpy_rotation(rpy, m);
    Matx33f R; for (int i = 0; i < 9; i++) R.val[i] = (float)m[i];
    return R;
}
Mat RPYAndPosition2transformationMatrix(Point3d rpy, Point3d position) {          // src/Plus.cpp:284-323
    double m[9]; rpy_rotation(rpy, m);
    Mat T = Mat::zeros(4, 4, CV_32FC1);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) T.at<float>(r, c) = (float)m[3 * r + c];
    T.at<float>(0, 3) = (float)position.x; T.at<float>(1, 3) = (float)position.y; T.at<float>(2, 3) = (float)position.z; T.at<float>(3, 3) = 1.0f;
    return T;
}
Matx33f transformationMatrix2rotationMatrix(Mat T) {                              // src/Plus.cpp:222-241
    Matx33f R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = T.at<float>(r, c);
    return R;
}
Point3d transformationMatrix2position(Mat T) { return Point3d(T.at<float>(0, 3), T.at<float>(1, 3), T.at<float>(2, 3)); }   // src/Plus.cpp:116-127

Frame::Frame() { obtainedGradients = false; obtainedGoodMatches = false; isKeyFrame = false; }   // src/Camera.cpp:6-11
Frame::~Frame() { grayImage.clear(); }

// ---------------------------------------------------------------- Matcher (src/Matcher.cpp)
Matcher::Matcher() { setMatcher(0); }
Matcher::Matcher(int _matcher) { setMatcher(_matcher); }
void Matcher::clear() {                                                          // :18-29
    keypoints_1.clear(); keypoints_2.clear(); descriptors_1.release(); descriptors_2.release();
    aux_matches1.clear(); aux_matches2.clear(); matches.clear(); goodMatches.clear(); sortedMatches.clear();
    slot1 = slot2 = -1;
}
void Matcher::setImageDimensions(int w, int h) { w_size = w; h_size = h; }        // :30-34
void Matcher::setKeypoints(vector<KeyPoint> a, vector<KeyPoint> b) { keypoints_1 = a; keypoints_2 = b; }   // :35-40
void Matcher::setDescriptors(Mat a, Mat b) { descriptors_1 = a; descriptors_2 = b; }                        // :42-46
void Matcher::setMatcher(int _matcher) {                                          // :49-78 (only Hamming is on the path)
    if (_matcher == USE_BRUTE_FORCE_HAMMING) cout << "Using Brute Force -Hamming CPU Matcher" << endl;
}
static void unpack_knn(const vector<vis_dmatch>& o, int n, vector<vector<DMatch> >& out) {
    out.assign((size_t)n, vector<DMatch>());
    for (int q = 0; q < n; q++)
        for (int k = 0; k < 2; k++) {
            const vis_dmatch& m = o[2 * (size_t)q + k];
            if (m.trainIdx < 0) continue;                 // OpenCV returns shorter vectors when fewer neighbours exist
            DMatch d; d.queryIdx = m.queryIdx; d.trainIdx = m.trainIdx; d.imgIdx = m.imgIdx; d.distance = m.distance;
            out[(size_t)q].push_back(d);
        }
}
void Matcher::computeMatches() {                                                  // :83-94 -> both knnMatch(k=2) calls
    vis_ctx* ctx = VisDevice::get();
    const int n1 = descriptors_1.rows, n2 = descriptors_2.rows;
    vector<vis_dmatch> o12(2 * (size_t)std::max(n1, 1)), o21(2 * (size_t)std::max(n2, 1));
    clock_t begin = clock();
    int rc = (slot1 >= 0 && slot2 >= 0) ? vis_bf_knn2_hamming(ctx, slot1, slot2, o12.data(), o21.data())
                                        : vis_bf_knn2_hamming_host(ctx, descriptors_1.data, n1, descriptors_2.data, n2, o12.data(), o21.data());
    if (rc) VisDevice::fail(rc, "computeMatches");
    clock_t knn2 = clock();
    unpack_knn(o12, n1, aux_matches1); unpack_knn(o21, n2, aux_matches2);
    // elapsed_*: the device time of the call (hipEvents on the context's stream, vis_last_timings) -- what the reference's clock()
    // around a blocking OpenCV-CUDA call measures; host clock() only if the context has no events
    vis_timings tm; const bool dev = vis_last_timings(ctx, &tm) == VIS_OK && tm.ms_knn > 0;
    elapsed_knn1 = elapsed_knn2 = dev ? 0.5e-3 * tm.ms_knn : 0.5 * double(knn2 - begin) / CLOCKS_PER_SEC;
}
void Matcher::computeBestMatches(int n_cells) {                                   // :353-367 (sym + sort + grid, fused on device)
    vis_ctx* ctx = VisDevice::get();
    vis_params p; vis_get_params(ctx, &p);
    if (p.n_cells != n_cells || p.w_size != w_size || p.h_size != h_size) {
        p.n_cells = n_cells; p.w_size = w_size; p.h_size = h_size;
        int rc = vis_set_params(ctx, &p);                 // matcher-side fields only: the device slots of the keyframes survive
        if (rc) VisDevice::fail(rc, "computeBestMatches/set_params");
    }
    const int n1 = (int)keypoints_1.size(), n2 = (int)keypoints_2.size();
    vector<vis_dmatch> good(1024), sym((size_t)std::max(n1, 1));
    int ng = 0, ns = 0, rc;
    clock_t begin = clock();
    if (slot1 >= 0 && slot2 >= 0) rc = vis_good_matches(ctx, slot1, slot2, good.data(), 1024, &ng, sym.data(), (int)sym.size(), &ns);
    else {
        vector<vis_dmatch> k12(2 * (size_t)std::max(n1, 1)), k21(2 * (size_t)std::max(n2, 1));
        auto pack = [](const vector<vector<DMatch> >& a, vector<vis_dmatch>& o) {
            for (size_t q = 0; q < a.size(); q++) for (int k = 0; k < 2; k++) {
                vis_dmatch& m = o[2 * q + k];
                if (k < (int)a[q].size()) { m.queryIdx = a[q][k].queryIdx; m.trainIdx = a[q][k].trainIdx; m.imgIdx = a[q][k].imgIdx; m.distance = a[q][k].distance; }
                else { m.queryIdx = (int)q; m.trainIdx = -1; m.imgIdx = -1; m.distance = 3.402823466e+38f; }
            }
        };
        pack(aux_matches1, k12); pack(aux_matches2, k21);
        rc = vis_good_matches_host(ctx, reinterpret_cast<const vis_keypoint*>(keypoints_1.data()), n1,
                                   reinterpret_cast<const vis_keypoint*>(keypoints_2.data()), n2, k12.data(), k21.data(),
                                   good.data(), 1024, &ng, sym.data(), (int)sym.size(), &ns);
    }
    if (rc) VisDevice::fail(rc, "computeBestMatches");
    clock_t best = clock();
    matches.clear(); goodMatches.clear(); sortedMatches.clear();
    for (int i = 0; i < ns; i++) matches.push_back(DMatch(sym[i].queryIdx, sym[i].trainIdx, sym[i].distance));
    for (int i = 0; i < ng; i++) goodMatches.push_back(DMatch(good[i].queryIdx, good[i].trainIdx, good[i].distance));
    nSymMatches = ns; nBestMatches = ng;
    vis_timings tm; const bool dev = vis_last_timings(ctx, &tm) == VIS_OK && tm.ms_filter > 0;
    elapsed_symMatches = elapsed_sortMatches = 0; elapsed_bestMatches = dev ? 1e-3 * tm.ms_filter : double(best - begin) / CLOCKS_PER_SEC;
}
void Matcher::getMatches(vector<KeyPoint>& a, vector<KeyPoint>& b) {              // :286-292
    for (unsigned i = 0; i < matches.size(); i++) { a.push_back(keypoints_1[matches[i].queryIdx]); b.push_back(keypoints_2[matches[i].trainIdx]); }
}
void Matcher::getGoodMatches(vector<KeyPoint>& a, vector<KeyPoint>& b) {          // :295-303
    a.clear(); b.clear();
    for (unsigned i = 0; i < goodMatches.size(); i++) { a.push_back(keypoints_1[goodMatches[i].queryIdx]); b.push_back(keypoints_2[goodMatches[i].trainIdx]); }
}
void Matcher::printStatistics() {                                                 // :369-382
    cout << "\nESTADISTICAS" << "\nNumero de matches simetricos: " << nSymMatches << "\tNumero de matches finales: " << nBestMatches << endl;
}

// ---------------------------------------------------------------- MatcherGPU (src/MatcherGPU.cpp)
MatcherGPU::MatcherGPU() { setGPUMatcher(0); }
MatcherGPU::MatcherGPU(int _matcher) { setGPUMatcher(_matcher); }
void MatcherGPU::setGPUFrames(Mat _frame1, Mat _frame2) { frameGPU1 = _frame1; frameGPU2 = _frame2; }   // include/MatcherGPU.hpp:20 (no body in the reference)
void MatcherGPU::setGPUMatcher(int _matcher) {                                    // :16-42
    matcherType = _matcher;
    matcherGPU = cuda::DescriptorMatcher::createBFMatcher(_matcher == USE_BRUTE_FORCE_GPU ? cuda::NORM_L2 : cuda::NORM_HAMMING);   // :23,:30
    if (_matcher == USE_BRUTE_FORCE_GPU_HAMMING) { cout << "Using Brute Force -Hamming GPU  Matcher" << endl; useGPU = true; }
    else if (_matcher == USE_BRUTE_FORCE_GPU) { cout << "L2 brute force is outside the hot path; using Hamming" << endl; useGPU = true; }
    else { useGPU = false; setMatcher(_matcher); }
}
void MatcherGPU::computeGPUMatches() {                                            // :44-66
    descriptorsGPU[0].release(); descriptorsGPU[1].release();
    descriptorsGPU[0].slot = slot1; descriptorsGPU[1].slot = slot2;               // "upload": descriptors are already resident
    computeMatches();
}

// ---------------------------------------------------------------- Camera / CameraGPU (src/Camera.cpp, src/CameraGPU.cpp)
Camera::Camera() {}
void Camera::Update(Mat _grayImage) {                                             // src/Camera.cpp:63-72
    currentFrame = new Frame();
    elapsed_computeGoodMatches = elapsed_descriptors = elapsed_detect = 0.0;
    _grayImage.copyTo(currentFrame->grayImage[0]);
    if (_grayImage.cols >= 16 && _grayImage.rows >= 16) {
        // resize(prev, next, Size(), 0.5, 0.5): the level sizes are cv::resize's cvRound(size * 0.5) (vis_half_pyramid_dims)
        int32_t lw[5], lh[5]; vis_half_pyramid_dims(_grayImage.cols, _grayImage.rows, lw, lh);
        uint8_t* lv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int i = 1; i < 5; i++) { currentFrame->grayImage[i].create(lh[i], lw[i], CV_8U); lv[i] = currentFrame->grayImage[i].data; }
        int rc = vis_camera_update(VisDevice::get(), _grayImage.data, _grayImage.cols, _grayImage.rows, (int)_grayImage.step, lv);
        if (rc) VisDevice::fail(rc, "Camera::Update");
    }
}
void Camera::computeGradient() {                                                  // src/Camera.cpp:167-184
    const Mat& img = currentFrame->grayImage[0];
    if (img.cols < 16 || img.rows < 16) return;
    int32_t lw[5], lh[5]; vis_half_pyramid_dims(img.cols, img.rows, lw, lh);           // Scharr(grayImage[lvl]): the size of that Mat
    int16_t* gx[5]; int16_t* gy[5]; uint8_t* g[5];
    for (int lvl = 0; lvl < 5; lvl++) {
        currentFrame->gradientX[lvl].create(lh[lvl], lw[lvl], CV_16S);
        currentFrame->gradientY[lvl].create(lh[lvl], lw[lvl], CV_16S);
        currentFrame->gradient[lvl].create(lh[lvl], lw[lvl], CV_8U);
        gx[lvl] = reinterpret_cast<int16_t*>(currentFrame->gradientX[lvl].data);
        gy[lvl] = reinterpret_cast<int16_t*>(currentFrame->gradientY[lvl].data);
        g[lvl] = currentFrame->gradient[lvl].data;
    }
    // Scharr(img, g, CV_16S, 1, 0, 3, 0, BORDER_DEFAULT): the "3" is cv::Scharr's scale argument
    int rc = vis_compute_gradient(VisDevice::get(), img.data, img.cols, img.rows, (int)img.step, 3, gx, gy, g);
    if (rc) VisDevice::fail(rc, "Camera::computeGradient");
    currentFrame->obtainedGradients = true;
}
static void patch_lists(Frame* last, bool patches, bool debug) {
    const vector<KeyPoint>& good = last->nextGoodMatches;
    const int cap = 200 * 121;
    vector<vector<float> > pb(5, vector<float>((size_t)cap * 4)), db(5, vector<float>((size_t)cap * 4));
    float* pp[5]; float* dp[5]; int np[5], nd[5];
    for (int l = 0; l < 5; l++) { pp[l] = pb[l].data(); dp[l] = db[l].data(); }
    int rc = vis_patch_points(VisDevice::get(), reinterpret_cast<const vis_keypoint*>(good.data()), (int)good.size(), cap, pp, np, dp, nd);
    if (rc) VisDevice::fail(rc, "Camera::ObtainPatchesPointsPreviousFrame");
    for (int l = 0; l < 5; l++) {
        // the reference push_backs 1x4 rows onto candidatePoints[lvl]: an N x 4 CV_32F matrix, appended call after call
        if (patches) {
            Mat old = last->candidatePoints[l], m;
            m.create(old.rows + np[l], 4, CV_32F);
            if (old.rows) std::memcpy(m.data, old.data, (size_t)old.rows * 16);
            if (np[l]) std::memcpy(m.data + (size_t)old.rows * 16, pp[l], (size_t)np[l] * 16);
            last->candidatePoints[l] = m;
        }
        if (debug) {
            Mat old = last->candidateDebugPoints[l], m;
            m.create(old.rows + nd[l], 4, CV_32F);
            if (old.rows) std::memcpy(m.data, old.data, (size_t)old.rows * 16);
            if (nd[l]) std::memcpy(m.data + (size_t)old.rows * 16, dp[l], (size_t)nd[l] * 16);
            last->candidateDebugPoints[l] = m;
        }
    }
}
void Camera::ObtainPatchesPointsPreviousFrame() { patch_lists(frameList[frameList.size() - 1], true, false); }   // :358-410
void Camera::ObtainDebugPointsPreviousFrame() { patch_lists(frameList[frameList.size() - 1], false, true); }     // :413-445
void Camera::saveFrame() {                                                        // :188-193
    currentFrame->isKeyFrame = true; frameList.push_back(currentFrame);
    if (currentFrame->gpuSlot == nextSlot) nextSlot = (nextSlot + 1) % 32;          // the keyframe keeps its device slot
}
void Camera::printStatistics() {                                                  // :325-356
    cout << "\nESTADISTICAS\tTdetect: " << elapsed_detect * 1000 << " ms\tTmatch: " << elapsed_computeGoodMatches * 1000
         << " ms\tNdetect: " << nPointsDetect << "\tNmatch: " << nBestMatches << endl;
}

CameraGPU::CameraGPU() {}
CameraGPU::CameraGPU(int d, int m, int w, int h, int c, int l) { initializateCameraGPU(d, m, w, h, c, l); }
void CameraGPU::initializateCameraGPU(int _detector, int _matcher, int _w_size, int _h_size, int _num_cells, int _length_path) {   // :21-42
    w_size[0] = _w_size; h_size[0] = _h_size;
    for (int lvl = 1; lvl < 5; lvl++) { w_size[lvl] = _w_size >> lvl; h_size[lvl] = _h_size >> lvl; }
    n_cells = _num_cells; w_patch = h_patch = _length_path;
    vis_ctx* ctx = VisDevice::get();
    vis_params p; vis_get_params(ctx, &p);
    p.nfeatures = 1000;                                   // cuda::ORB::create(1000), src/CameraGPU.cpp:99
    p.n_cells = _num_cells; p.w_size = _w_size; p.h_size = _h_size;
    int rc = vis_set_params(ctx, &p);
    if (rc) VisDevice::fail(rc, "initializateCameraGPU");
    setGPUDetector(_detector);
    setGPUMatcher(_matcher);
    num_images = 0;
}
void CameraGPU::setGPUDetector(int _detector) {                                   // :44-69
    if (_detector == USE_ORB) { useGPU = true; detectorType = _detector; cout << "Using ORB detector in GPU" << endl; }
    else { useGPU = true; detectorType = USE_ORB; cout << "Only ORB is on the hot path: using ORB detector in GPU" << endl; }
}
void CameraGPU::detectGPUFeatures() { nPointsDetect = detectAndComputeGPUFeatures(); }   // include/CameraGPU.hpp:22 (no body in the reference)
void CameraGPU::setGPUMatcher(int _matcher) { matcherGPU.setGPUMatcher(_matcher); matcherGPU.setImageDimensions(w_size[0], h_size[0]); }   // :119-123
int CameraGPU::detectAndComputeGPUFeatures() {                                    // :71-117
    keypointsGPU.release(); descriptorsGPU.release(); frameGPU.release();
    vis_ctx* ctx = VisDevice::get();
    const Mat& img = currentFrame->grayImage[0];
    const int cap = 4096;
    vector<vis_keypoint> kps(cap);
    currentFrame->descriptors.create(cap, 32, CV_8U);
    int n = 0;
    const int slot = nextSlot;                            // kept only if saveFrame() makes this frame a keyframe
    int rc = vis_orb_detect_compute(ctx, img.data, img.cols, img.rows, (int)img.step, slot, kps.data(), currentFrame->descriptors.data, cap, &n);
    if (rc) VisDevice::fail(rc, "detectAndComputeGPUFeatures");
    currentFrame->gpuSlot = slot; currentFrame->gpuGen = ++slotGen[slot];
    frameGPU.slot = keypointsGPU.slot = descriptorsGPU.slot = slot;
    currentFrame->keypoints.resize((size_t)n);
    if (n) std::memcpy(static_cast<void*>(currentFrame->keypoints.data()), kps.data(), (size_t)n * sizeof(vis_keypoint));
    currentFrame->descriptors = currentFrame->descriptors.rowRange(0, n);
    nPointsDetect = n;
    return nPointsDetect;
}
void CameraGPU::computeGPUGoodMatches() {                                         // :125-136
    Frame* last = frameList[frameList.size() - 1];
    matcherGPU.clear();
    matcherGPU.setKeypoints(last->keypoints, currentFrame->keypoints);
    matcherGPU.setDescriptors(last->descriptors, currentFrame->descriptors);
    // device-resident descriptors of the last keyframe, unless its slot has been recycled since (more than 32 live keyframes):
    // then the host copies the reference keeps in frameList.back() are matched instead (src/CameraGPU.cpp:129-130)
    const bool resident = last->gpuSlot >= 0 && slotGen[last->gpuSlot] == last->gpuGen && last->gpuSlot != currentFrame->gpuSlot;
    matcherGPU.slot1 = resident ? last->gpuSlot : -1; matcherGPU.slot2 = resident ? currentFrame->gpuSlot : -1;
    matcherGPU.computeGPUMatches();
    matcherGPU.computeBestMatches(n_cells);
    matcherGPU.getGoodMatches(last->nextGoodMatches, currentFrame->prevGoodMatches);
    currentFrame->obtainedGoodMatches = true;
}
bool CameraGPU::addGPUKeyframe() {                                                // :138-202
    clock_t cbegin = clock();
    nPointsDetect = detectAndComputeGPUFeatures();
    clock_t cdetect = clock();
    vis_timings tdet; const bool dev_det = vis_last_timings(VisDevice::get(), &tdet) == VIS_OK && tdet.ms_total > 0;
    if ((nPointsDetect > 1) && (frameList.size() != 0)) {
        computeGPUGoodMatches();
        clock_t cgood = clock();
        computeGradient();
        clock_t cgradient = clock();
        ObtainPatchesPointsPreviousFrame();
        ObtainDebugPointsPreviousFrame();
        clock_t cpatches = clock();
        saveFrame();
        nBestMatches = (int)matcherGPU.goodMatches.size();
        elapsed_detect = dev_det ? 1e-3 * tdet.ms_total : double(cdetect - cbegin) / CLOCKS_PER_SEC;   // device time of the detect chain
        const double dev_match = matcherGPU.elapsed_knn1 + matcherGPU.elapsed_knn2 + matcherGPU.elapsed_bestMatches;
        elapsed_computeGoodMatches = dev_match > 0 ? dev_match : double(cgood - cdetect) / CLOCKS_PER_SEC;
        elapsed_computeGradient = double(cgradient - cgood) / CLOCKS_PER_SEC;
        elapsed_computePatches = double(cpatches - cgradient) / CLOCKS_PER_SEC;
        elapsed_detect_sum += elapsed_detect; elapsed_computeGoodMatches_sum += elapsed_computeGoodMatches;
        nPointsDetect_sum += nPointsDetect; nBestMatches_sum += nBestMatches;
        const double nn = double(frameList.size() - 1);
        elapsed_detect_mean = elapsed_detect_sum / nn; elapsed_computeGoodMatches_mean = elapsed_computeGoodMatches_sum / nn;
        nPointsDetect_mean = nPointsDetect_sum / nn; nBestMatches_mean = nBestMatches_sum / nn;
    } else if ((nPointsDetect > 1) && (frameList.size() == 0)) {
        computeGradient();
        saveFrame();
        cout << "First Image detected" << "list = " << frameList.size() << endl;
    }
    return currentFrame->isKeyFrame;                      // SPEC: the reference function has no return statement
}

// ---------------------------------------------------------------- ImageReader (src/ImageReader.cpp)
ImageReader::ImageReader() { setPath(""); TimeStep = 0.0; }
ImageReader::ImageReader(string _directory) { setPath(_directory); searchImages(); computeTimeStep(); }   // :9-14
void ImageReader::setPath(string _directory) { path = _directory; if (!path.empty() && path.back() != '/') path += '/'; }
void ImageReader::setRawSize(int w, int h) { raw_w = w; raw_h = h; }
string ImageReader::getImageName(int index) {                                     // :22-39: file name up to the first '.'
    string n = file_names[(size_t)index];
    const size_t slash = n.rfind('/');
    if (slash != string::npos) n = n.substr(slash + 1);
    const size_t dot = n.find('.');
    return (dot != string::npos && dot != 0) ? n.substr(0, dot) : n;
}
long int ImageReader::getImageTime(int index) { return vis_image_time(file_names[(size_t)index].c_str()); }   // :41-47
void ImageReader::searchImages() {                                                // :49-74
    cout << "Searching images files in directory ... ";
    int count = 0;
    int rc = vis_image_list(path.c_str(), nullptr, 0, &count);
    if (rc) { cout << "Could not open directory of images: " << path << endl << "Exiting..." << endl; exit(0); }
    vector<char> buf((size_t)count * 300 + 1);
    rc = vis_image_list(path.c_str(), buf.data(), (int)buf.size(), &count);
    if (rc) VisDevice::fail(rc, "ImageReader::searchImages");
    file_names.clear();
    for (char* s = buf.data(); *s;) { char* e = std::strchr(s, '\n'); if (!e) break; file_names.push_back(path + string(s, e)); s = e + 1; }
    if (file_names.size() < 15) { cout << "\nInsufficient number of images found. Please use a larger dataset" << endl << "Exiting..." << endl; exit(0); }
    cout << file_names.size() << " found" << endl;
}
Mat ImageReader::getImage(int index) {                                            // :80-82
    const string& f = file_names[(size_t)index];
    int w = raw_w, h = raw_h;
    const bool raw = f.size() > 4 && f.compare(f.size() - 4, 4, ".raw") == 0;
    if (!raw && vis_image_info(f.c_str(), &w, &h) != 0) return Mat();             // (PGM or greyscale PNG by its magic bytes) imread returns an empty Mat on failure
    if (w < 1 || h < 1) return Mat();
    Mat m(h, w, CV_8U);
    if (vis_image_read(f.c_str(), m.data, (int)m.step, w, h) != 0) return Mat();
    return m;
}
size_t ImageReader::getSize() { return file_names.size(); }
void ImageReader::computeTimeStep() { TimeStep = (double)getImageTime(1) - (double)getImageTime(0); }   // :107-112

// ---------------------------------------------------------------- vi::CameraModel (src/CameraModel.cpp)
namespace vi {
// the calibration files are cv::FileStorage XML (calibration/calibrationEUROC.xml): <tag type_id=...> value </tag>, matrices
// carry their numbers in a <data> element.  This reader handles exactly that shape; no OpenCV.
static bool xml_element(const string& doc, const string& tag, string& out) {
    size_t a = doc.find("<" + tag);
    while (a != string::npos) {                              // "<tag" followed by '>' or whitespace (not a longer tag name)
        const char c = doc[a + tag.size() + 1];
        if (c == '>' || c == ' ' || c == '\t' || c == '\n' || c == '\r') break;
        a = doc.find("<" + tag, a + 1);
    }
    if (a == string::npos) return false;
    const size_t b = doc.find('>', a), e = doc.find("</" + tag + ">", a);
    if (b == string::npos || e == string::npos || e < b) return false;
    out = doc.substr(b + 1, e - b - 1);
    return true;
}
static vector<double> xml_numbers(const string& doc, const string& tag) {
    vector<double> v; string el;
    if (!xml_element(doc, tag, el)) return v;
    string data;
    if (xml_element(el, "data", data)) el = data;
    std::istringstream is(el); double x;
    while (is >> x) v.push_back(x);
    return v;
}
Mat optimal_new_camera_matrix_alpha1(const float K[4], const float dist[4], int in_w, int in_h, int out_w, int out_h);
void CameraModel::GetCameraModel(string _calibration_path) {                          // src/CameraModel.cpp:16-101
    valid_ = true;
    std::ifstream f(_calibration_path.c_str());
    if (!f.is_open()) {
        cout << " ... not found" << endl << "Cannot operate without calibration" << endl << "Exiting..." << endl;
        valid_ = false; exit(0);                                                      // :46-52
    }
    cout << " ... found" << endl;
    std::stringstream ss; ss << f.rdbuf();
    const string doc = ss.str();
    auto num = [&](const char* tag, double dflt) { vector<double> v = xml_numbers(doc, tag); return v.empty() ? dflt : v[0]; };
    in_width_ = (int)num("in_width", 0); in_height_ = (int)num("in_height", 0);
    out_width_ = (int)num("out_width", 0); out_height_ = (int)num("out_height", 0);
    vector<double> cal = xml_numbers(doc, "calibration_values"), dist = xml_numbers(doc, "rectification"), imu = xml_numbers(doc, "imu2cam0Transformation");
    imu2cam0Transformation = Mat::eye(4, 4, CV_32FC1);
    for (size_t i = 0; i < imu.size() && i < 16; i++) imu2cam0Transformation.at<float>((int)i / 4, (int)i % 4) = (float)imu[i];
    camera_frecuency = (float)num("camera_frecuency", 0); imu_frecuency = (float)num("imu_frecuency", 0);
    min_features = (int)num("min_features", 0); num_max_keyframes = (int)num("num_max_keyframes", 0); start_index = (int)num("start_index", 0);
    use_gt = (int)num("use_gt", 0); use_ros = (int)num("use_ros", 0); num_cells = (int)num("num_cells", 0); length_patch = (int)num("length_patch", 0);
    detector = (int)num("detector", 0); matcher = (int)num("matcher", 0);
    for (int i = 0; i < 4; i++) { input_calibration_[i] = i < (int)cal.size() ? (float)cal[i] : 0.f; dist_coeffs_[i] = i < (int)dist.size() ? (float)dist[i] : 0.f; }
    if (input_calibration_[2] < 1 && input_calibration_[3] < 1) {                     // relative intrinsics, :60-68
        cout << "WARNING: cx = " << input_calibration_[2] << " < 1, which should not be the case for normal cameras" << endl;
        input_calibration_[0] *= in_width_; input_calibration_[1] *= in_height_; input_calibration_[2] *= in_width_; input_calibration_[3] *= in_height_;
    }
    original_intrinsic_camera_ = Mat::zeros(3, 3, CV_32FC1);
    original_intrinsic_camera_.at<float>(0, 0) = input_calibration_[0]; original_intrinsic_camera_.at<float>(1, 1) = input_calibration_[1];
    original_intrinsic_camera_.at<float>(0, 2) = input_calibration_[2]; original_intrinsic_camera_.at<float>(1, 2) = input_calibration_[3];
    original_intrinsic_camera_.at<float>(2, 2) = 1;
    if (dist_coeffs_[0] == 0) {                                                       // :78-83
        cout << "Distortion coefficients not found ... not rectifying" << endl;
        valid_ = false;
        output_intrinsic_camera_ = original_intrinsic_camera_;
    } else {
        // :84-88: K_ = getOptimalNewCameraMatrix(K, dist, Size(in), alpha = 1.0, Size(out), nullptr, false).  The reference never
        // remaps the frames of the GPU main (VISystemGPU::AddFrameGPU hands the raw image on), but every intrinsic it uses
        // afterwards -- InitializePyramid, EstimatePoseFeatures, findEssentialMat -- is this matrix, so it is restated here.
        cout << "Distortion coefficients found ... rectifying" << endl;
        output_intrinsic_camera_ = optimal_new_camera_matrix_alpha1(input_calibration_, dist_coeffs_, in_width_, in_height_, out_width_, out_height_);
    }
}

// cv::getOptimalNewCameraMatrix(alpha = 1, centerPrincipalPoint = false) as calib3d 3.2 computes it (calibration.cpp:
// cvGetOptimalNewCameraMatrix -> icvGetRectangles -> cvUndistortPoints) -- written from the published algorithm, OpenCV is not
// available here: PARITY UNPINNED like the rest of the OpenCV-owned arithmetic.
//   1. a 9 x 9 grid of pixel positions (x * w / 8, y * h / 8), stored as float;
//   2. each is undistorted into normalised coordinates by 5 fixed-point iterations of the inverse Brown model (double);
//   3. outer = bounding box of the 81 results (float); with alpha = 1 the new projection maps it onto the output viewport:
//      fx' = (out_w - 1) / outer.width, cx' = -fx' * outer.x (same for y).
Mat optimal_new_camera_matrix_alpha1(const float K[4], const float dist[4], int in_w, int in_h, int out_w, int out_h) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    float oX0 = FLT_MAX, oX1 = -FLT_MAX, oY0 = FLT_MAX, oY1 = -FLT_MAX;
    const int N = 9;
    for (int y = 0; y < N; y++)
        for (int x = 0; x < N; x++) {
            const float u = (float)x * in_w / (N - 1), v = (float)y * in_h / (N - 1);
            double xn = ((double)u - cx) * (1.0 / fx), yn = ((double)v - cy) * (1.0 / fy);
            const double x0 = xn, y0 = yn;
            for (int j = 0; j < 5; j++) {
                const double r2 = xn * xn + yn * yn;
                const double icdist = 1.0 / (1 + (k2 * r2 + k1) * r2);
                const double dX = 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn);
                const double dY = p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn;
                xn = (x0 - dX) * icdist; yn = (y0 - dY) * icdist;
            }
            const float px = (float)xn, py = (float)yn;
            oX0 = std::min(oX0, px); oX1 = std::max(oX1, px); oY0 = std::min(oY0, py); oY1 = std::max(oY1, py);
        }
    const float ow = oX1 - oX0, oh = oY1 - oY0;                                       // cv::Rect_<float>(oX0, oY0, oX1 - oX0, oY1 - oY0)
    const double fx1 = (out_w - 1) / (double)ow, fy1 = (out_h - 1) / (double)oh;
    const double cx1 = -fx1 * oX0, cy1 = -fy1 * oY0;
    Mat M = Mat::zeros(3, 3, CV_32FC1);
    M.at<float>(0, 0) = (float)fx1; M.at<float>(1, 1) = (float)fy1; M.at<float>(0, 2) = (float)cx1; M.at<float>(1, 2) = (float)cy1; M.at<float>(2, 2) = 1;
    return M;
}

// initUndistortRectifyMap(K, dist, Mat(), K', Size(out)) maps output pixel (u, v) to the source position
//   x = (u - cx') / fx', y = (v - cy') / fy';  r2 = x^2 + y^2;  kr = 1 + k1 r2 + k2 r2^2
//   (fx (x kr + 2 p1 x y + p2 (r2 + 2 x^2)) + cx,  fy (y kr + p1 (r2 + 2 y^2) + 2 p2 x y) + cy)
// and remap(INTER_LINEAR, BORDER_CONSTANT 0) leaves a pixel black when no tap of its 2 x 2 footprint lies inside the source.
void CameraModel::RectifiedROI(int roi[4]) const {
    const double fx = input_calibration_[0], fy = input_calibration_[1], cx = input_calibration_[2], cy = input_calibration_[3];
    const double k1 = dist_coeffs_[0], k2 = dist_coeffs_[1], p1 = dist_coeffs_[2], p2 = dist_coeffs_[3];
    const double fxn = output_intrinsic_camera_.at<float>(0, 0), fyn = output_intrinsic_camera_.at<float>(1, 1);
    const double cxn = output_intrinsic_camera_.at<float>(0, 2), cyn = output_intrinsic_camera_.at<float>(1, 2);
    auto inside = [&](int u, int v) {
        const double x = (u - cxn) / fxn, y = (v - cyn) / fyn, r2 = x * x + y * y, kr = 1 + k1 * r2 + k2 * r2 * r2;
        const double su = fx * (x * kr + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)) + cx;
        const double sv = fy * (y * kr + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y) + cy;
        return su > -1.0 && su < in_width_ && sv > -1.0 && sv < in_height_;
    };
    const int x_middle = (int)((out_width_ - 1) * 0.5), y_middle = (int)((out_height_ - 1) * 0.5);   // :170-171
    int x1 = 0, y1 = 0, x2 = out_width_ - 1, y2 = out_height_ - 1;
    while (x1 < x2 && !inside(x1, y_middle)) x1++;
    while (x2 > x1 && !inside(x2, y_middle)) x2--;
    while (y1 < y2 && !inside(x_middle, y1)) y1++;
    while (y2 > y1 && !inside(x_middle, y2)) y2--;
    roi[0] = x1 + 5; roi[1] = y1 + 5; roi[2] = x2 - 5; roi[3] = y2 - 5;                // :193-197
}

// ---------------------------------------------------------------- vi::VISystem / vi::VISystemGPU (src/VISystem.cpp, src/VISystemGPU.cpp)
VISystem::VISystem() {
    for (int i = 0; i < 9; i++) ransacR[i] = (i % 4 == 0) ? 1.f : 0.f;
    ransacT[0] = ransacT[1] = ransacT[2] = 0.f;
    std::memset(&lastAlignment, 0, sizeof(lastAlignment));
}
void VISystem::Calibration(string _calibration_path) {                                // src/VISystem.cpp:208-221
    cout << "Reading calibration xml file";
    camera_model = new CameraModel();
    camera_model->GetCameraModel(_calibration_path);
    w = camera_model->GetOutputWidth();
    h = camera_model->GetOutputHeight();
    if (w % 2 != 0 || h % 2 != 0) {
        cout << "Output image dimensions must be multiples of 32. Choose another output dimentions" << endl << "Exiting..." << endl;
        exit(0);
    }
}
void VISystem::InitializePyramid(int _width, int _height, Mat _K) {                    // :1451-1493
    w_[0] = _width; h_[0] = _height; K_[0] = _K;
    fx_[0] = _K.at<float>(0, 0); fy_[0] = _K.at<float>(1, 1); cx_[0] = _K.at<float>(0, 2); cy_[0] = _K.at<float>(1, 2);
    invfx_[0] = 1 / fx_[0]; invfy_[0] = 1 / fy_[0]; invcx_[0] = 1 / cx_[0]; invcy_[0] = 1 / cy_[0];
    for (int lvl = 1; lvl < 5; lvl++) {
        w_[lvl] = _width >> lvl; h_[lvl] = _height >> lvl;
        fx_[lvl] = fx_[lvl - 1] * 0.5; fy_[lvl] = fy_[lvl - 1] * 0.5;
        cx_[lvl] = (cx_[0] + 0.5) / ((int)1 << lvl) - 0.5; cy_[lvl] = (cy_[0] + 0.5) / ((int)1 << lvl) - 0.5;
        K_[lvl] = Mat::eye(3, 3, CV_32FC1);
        K_[lvl].at<float>(0, 0) = fx_[lvl]; K_[lvl].at<float>(1, 1) = fy_[lvl]; K_[lvl].at<float>(0, 2) = cx_[lvl]; K_[lvl].at<float>(1, 2) = cy_[lvl];
        invfx_[lvl] = 1 / fx_[lvl]; invfy_[lvl] = 1 / fy_[lvl]; invcx_[lvl] = 1 / cx_[lvl]; invcy_[lvl] = 1 / cy_[lvl];
    }
}
void VISystem::setGtRes(Mat TranslationResGT, Mat RotationResGT) {                     // :415-419
    TranslationResidual = TranslationResGT;
    Matx33f R; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R(r, c) = RotationResGT.at<float>(r, c);
    RotationResidual = RPY2rotationMatrix(rotationMatrix2RPY(R));
}
// Gauss-Newton photometric alignment of the previous keyframe's candidate points to the current frame, :1113-1448.
// Options (:1115-1121) are the defaults of vis_align_params.  The reference seeds the pose with the IMU's residual rotation and
// the ground-truth translation residual (:1133-1166); the IMU core is outside this build (identity), the translation seed is
// whatever setGtRes stored (zero without ground truth).  GUI calls (imshow / waitKey, :1250-1263) are dropped.
void VISystem::EstimatePoseFeatures(Frame* _previous_frame, Frame* _current_frame) {
    vis_align_params ap; vis_default_align_params(&ap);
    ap.fx = fx_[0]; ap.fy = fy_[0]; ap.cx = cx_[0]; ap.cy = cy_[0];
    const uint8_t* g1[5]; const uint8_t* g2[5]; const int16_t* gx[5]; const int16_t* gy[5]; const float* cd[5]; int32_t n[5];
    // With a rectifying calibration the system's (w, h) is the ROI of src/VISystem.cpp:162-205 while the frames keep their
    // size: the reference bounds-checks against w_[lvl], h_[lvl] and indexes the full-size Mats (:1267-1305).  The C ABI takes
    // dense levels of the sizes vis_half_pyramid_dims(w_[0], h_[0]) gives (w_[lvl] x h_[lvl], or one column / row more where a size
    // does not halve exactly), so the top-left window of that size of every level is what is handed over.
    int32_t aw[5], ah[5]; vis_half_pyramid_dims(w_[0], h_[0], aw, ah);
    Mat crop[4][5];
    auto window = [&](const Mat& m, int l, Mat& keep) -> const uint8_t* {
        if (m.empty() || (m.cols == aw[l] && m.rows == ah[l] && m.step == (size_t)m.cols * m.elemSize())) return m.data;
        if (m.cols < aw[l] || m.rows < ah[l]) return nullptr;
        keep.create(ah[l], aw[l], m.depth);
        for (int y = 0; y < ah[l]; y++) std::memcpy(keep.data + (size_t)y * keep.step, m.data + (size_t)y * m.step, keep.step);
        return keep.data;
    };
    for (int l = 0; l < 5; l++) {
        g1[l] = window(_previous_frame->grayImage[l], l, crop[0][l]); g2[l] = window(_current_frame->grayImage[l], l, crop[1][l]);
        gx[l] = reinterpret_cast<const int16_t*>(window(_previous_frame->gradientX[l], l, crop[2][l]));
        gy[l] = reinterpret_cast<const int16_t*>(window(_previous_frame->gradientY[l], l, crop[3][l]));
        cd[l] = reinterpret_cast<const float*>(_previous_frame->candidatePoints[l].data);
        n[l] = _previous_frame->candidatePoints[l].rows;
        if (!g1[l] || !g2[l] || !gx[l] || !gy[l]) n[l] = 0;                            // Update() could not build the half pyramid (frame smaller than 16 x 16)
    }
    const float sx = TranslationResidual.at<float>(0, 0), sy = TranslationResidual.at<float>(1, 0), sz = TranslationResidual.at<float>(2, 0);
    const SE3 current_pose(Matx33f::eye(), SE3::Point(-sx, -sy, -sz));                 // :1159
    int rc = vis_estimate_pose_features(VisDevice::get(), &ap, w_[0], h_[0], g1, g2, gx, gy, cd, n, &current_pose.v, &lastAlignment);
    if (rc) VisDevice::fail(rc, "EstimatePoseFeatures");
    _previous_frame->rigid_transformation_.v = lastAlignment.pose;                     // :1445
}
int VISystem::EstimatePoseFeaturesRansac(Frame* prev, Frame* cur) {                    // :1655-1708
    vis_ctx* ctx = VisDevice::get();
    const int m = (int)prev->nextGoodMatches.size();
    vector<float> p1(2 * (size_t)std::max(m, 1)), p2(2 * (size_t)std::max(m, 1));
    for (int i = 0; i < m; i++) {                                                     // KeyPoint::convert, :1673-1674
        p1[2 * i] = prev->nextGoodMatches[i].pt.x; p1[2 * i + 1] = prev->nextGoodMatches[i].pt.y;
        p2[2 * i] = cur->prevGoodMatches[i].pt.x; p2[2 * i + 1] = cur->prevGoodMatches[i].pt.y;
    }
    double E[9], R[9], t[3]; int ninl = 0, iters = 0, ngood = 0;
    int rc = vis_essential_ransac(ctx, p1.data(), p2.data(), m, E, nullptr, &ninl, &iters);   // :1679-1680
    if (rc) VisDevice::fail(rc, "findEssentialMat");
    for (int i = 0; i < 9; i++) ransacR[i] = (i % 4 == 0) ? 1.f : 0.f;
    ransacT[0] = ransacT[1] = ransacT[2] = 0.f;
    lastInliers = ninl; lastPoseGood = 0;
    if (ninl > 0) {
        rc = vis_recover_pose(ctx, E, p1.data(), p2.data(), m, R, t, &ngood);          // :1701
        if (rc) VisDevice::fail(rc, "recoverPose");
        for (int i = 0; i < 9; i++) ransacR[i] = (float)R[i];                         // convertTo(CV_32FC1), :1702-1703
        for (int i = 0; i < 3; i++) ransacT[i] = (float)t[i];
        lastPoseGood = ngood;
    }
    return ninl;
}
// :1567-1635.  RotationResCam / translationResEst are the residual camera motion of the step (AddFrameGPU sets them from
// the alignment result; in the reference they come from the IMU / F2FRansac path of the CPU main).  The IMU pose chain
// composes the identity: imuCore is outside this build.
void VISystem::Track() {
    SE3::Point tRes(translationResEst.x, translationResEst.y, translationResEst.z);
    current_poseCam = SE3(RotationResCam, tRes);
    final_poseCam = final_poseCam * current_poseCam;
    const SE3::Point t = final_poseCam.translation();
    positionCam.x = t(0); positionCam.y = t(1); positionCam.z = t(2);
    qOrientationCam.x = final_poseCam.unit_quaternion().x(); qOrientationCam.y = final_poseCam.unit_quaternion().y();
    qOrientationCam.z = final_poseCam.unit_quaternion().z(); qOrientationCam.w = final_poseCam.unit_quaternion().w();
    RPYOrientationCam = toRPY(qOrientationCam);
    prev_world2camTransformation = RPYAndPosition2transformationMatrix(RPYOrientationCam, positionCam);
    current_poseImu = SE3(Matx33f::eye(), SE3::Point(0.0, 0.0, 0.0));
    final_poseImu = final_poseImu * current_poseImu;
    const SE3::Point t2 = final_poseImu.translation();
    positionImu.x = -t2(0); positionImu.y = -t2(2); positionImu.z = -t2(1);
    qOrientationImu.x = final_poseImu.unit_quaternion().x(); qOrientationImu.y = final_poseImu.unit_quaternion().y();
    qOrientationImu.z = final_poseImu.unit_quaternion().z(); qOrientationImu.w = final_poseImu.unit_quaternion().w();
    RPYOrientationImu = toRPY(qOrientationImu);
}

VISystemGPU::VISystemGPU() { initialized = false; distortion_valid = false; depth_available = false; num_keyframes = 0; }
VISystemGPU::VISystemGPU(int, char*[]) { initialized = false; distortion_valid = false; depth_available = false; num_keyframes = 0; }   // ros::init dropped (out of scope)
VISystemGPU::~VISystemGPU() { cout << "SLAM System shutdown ..." << endl; }
void VISystemGPU::InitializeSystemGPU(string _calPath, Point3d _iniPosition, Point3d _iniVelocity, Point3d _iniRPY, Mat image) {   // src/VISystemGPU.cpp:39-129
    currentImage = image;
    Calibration(_calPath);
    imu2camTransformation = camera_model->imu2cam0Transformation;
    imu2camRotation = transformationMatrix2rotationMatrix(imu2camTransformation);
    imu2camTranslation = transformationMatrix2position(imu2camTransformation);
    K = camera_model->GetK();
    w_input = camera_model->GetInputWidth(); h_input = camera_model->GetInputHeight();
    map1 = camera_model->GetMap1(); map2 = camera_model->GetMap2();
    fx = K.at<float>(0, 0); fy = K.at<float>(1, 1); cx = K.at<float>(0, 2); cy = K.at<float>(1, 2);
    distortion_valid = camera_model->IsValid();
    if (distortion_valid) {                                                          // :66-75, CalculateROI (src/VISystem.cpp:162-205)
        int roi[4]; camera_model->RectifiedROI(roi);
        w = roi[2] - roi[0]; h = roi[3] - roi[1];                                     // :201-204
        cout << "distortion detected" << endl;
        cout << "Input width = " << w_input << "\t" << " Output width = " << w << endl;
        cout << "Input height = " << h_input << "\t" << " Output height = " << h << endl;
    } else { w = w_input; h = h_input; }
    InitializePyramid(w, h, K);
    initialized = true;
    cout << "Initializing system ... done" << endl << endl;
    // initial IMU pose, :97-104
    positionImu = _iniPosition; velocityImu = _iniVelocity; RPYOrientationImu = _iniRPY;
    qOrientationImu = toQuaternion(_iniRPY.x, _iniRPY.y, _iniRPY.z);
    world2imuTransformation = RPYAndPosition2transformationMatrix(RPYOrientationImu, positionImu);
    world2imuRotation = transformationMatrix2rotationMatrix(world2imuTransformation);
    final_poseImu = SE3(SE3::Quaternion((float)qOrientationImu.w, (float)qOrientationImu.x, (float)qOrientationImu.y, (float)qOrientationImu.z), SE3::Point(0.0, 0.0, 0.0));
    // initial camera pose, :107-113: positionCam = imu2camTransformation * (positionImu, 1), float like the Mat product
    const float pi[4] = {(float)positionImu.x, (float)positionImu.y, (float)positionImu.z, 1.0f};
    float pc[3], vc[3];
    for (int r = 0; r < 3; r++) {
        double acc = 0; for (int c = 0; c < 4; c++) acc += (double)imu2camTransformation.at<float>(r, c) * (double)pi[c];
        pc[r] = (float)acc;
        const float vi3[3] = {(float)velocityImu.x, (float)velocityImu.y, (float)velocityImu.z};
        double av = 0; for (int c = 0; c < 3; c++) av += (double)imu2camRotation(r, c) * (double)vi3[c];
        vc[r] = (float)av;
    }
    positionCam = Point3d(pc[0], pc[1], pc[2]);
    velocityCam = Point3d(vc[0], vc[1], vc[2]);
    RPYOrientationCam = rotationMatrix2RPY(imu2camRotation * world2imuRotation);
    qOrientationCam = toQuaternion(RPYOrientationCam.x, RPYOrientationCam.y, RPYOrientationCam.z);
    final_poseCam = SE3(SE3::Quaternion((float)qOrientationCam.w, (float)qOrientationCam.x, (float)qOrientationCam.y, (float)qOrientationCam.z),
                        SE3::Point((float)-positionCam.x, (float)-positionCam.z, (float)-positionCam.y));
    // imuCore.createPublisher / initializate / setImuInitialVelocity (:116-118): the ROS IMU core is outside this build
    num_max_keyframes = camera_model->min_features;                                  // :121 (sic: the reference stores min_features)
    min_features = camera_model->min_features;
    start_index = camera_model->start_index;
    InitializeCameraGPU(camera_model->detector, camera_model->matcher, w, h, camera_model->num_cells, camera_model->length_patch);
    // intrinsics of the essential-matrix path: findEssentialMat(focal = fx, pp = (cx, cy)), src/VISystem.cpp:1679
    vis_ctx* ctx = VisDevice::get();
    vis_params p; vis_get_params(ctx, &p);
    p.fx = fx; p.fy = fx; p.cx = cx; p.cy = cy;
    int rc = vis_set_params(ctx, &p);
    if (rc) VisDevice::fail(rc, "InitializeSystemGPU");
}
void VISystemGPU::InitializeCameraGPU(int d, int m, int w_, int h_, int c, int l) { cameraGPU.initializateCameraGPU(d, m, w_, h_, c, l); }   // :132-135
void VISystemGPU::AddFrameGPU(Mat _currentImage, vector<Point3d> _imuAngularVelocity, vector<Point3d> _imuAcceleration) {   // :137-175
    prevImage = currentImage;
    currentImage = _currentImage.clone();
    (void)_imuAngularVelocity; (void)_imuAcceleration;    // imuCore.setImuData / estimate (:142-143): the ROS IMU core is outside this build
    cameraGPU.Update(_currentImage);
    bool key_added = cameraGPU.addGPUKeyframe();
    (void)key_added;
    num_keyframes = (int)cameraGPU.frameList.size();
    if (cameraGPU.frameList.size() > 1) {
        cameraGPU.printStatistics();
        if (num_keyframes > num_max_keyframes) FreeLastFrameGPU();
        Frame* prev = cameraGPU.frameList[cameraGPU.frameList.size() - 2];
        EstimatePoseFeatures(prev, cameraGPU.frameList[cameraGPU.frameList.size() - 1]);
        // SPEC: the residual camera motion Track() composes is the alignment estimate (the commented line at
        // src/VISystem.cpp:1580 shows that intent; the members are otherwise only set by the CPU main's IMU / F2F path)
        RotationResCam = prev->rigid_transformation_.rotationMatrix();
        const SE3::Point tt = prev->rigid_transformation_.translation();
        translationResEst = Point3f(tt(0), tt(1), tt(2));
        Track();
    }
}
void VISystemGPU::FreeLastFrameGPU() {                                                // :178-182 (without the reference's leak)
    delete cameraGPU.frameList[0];
    cameraGPU.frameList.erase(cameraGPU.frameList.begin());
}
}  // namespace vi
