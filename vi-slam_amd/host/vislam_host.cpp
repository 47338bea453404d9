// vislam_host.cpp -- adapter bodies; every method cites the reference body it mirrors.
#include "vislam_host.hpp"
#include <cstdlib>
#include <ctime>

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
    elapsed_knn1 = elapsed_knn2 = 0.5 * double(knn2 - begin) / CLOCKS_PER_SEC;
}
void Matcher::computeBestMatches(int n_cells) {                                   // :353-367 (sym + sort + grid, fused on device)
    vis_ctx* ctx = VisDevice::get();
    vis_params p; vis_get_params(ctx, &p);
    if (p.n_cells != n_cells || p.w_size != w_size || p.h_size != h_size) {
        p.n_cells = n_cells; p.w_size = w_size; p.h_size = h_size;
        int rc = vis_set_params(ctx, &p);                 // note: drops device slots; callers re-detect (only at init)
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
    elapsed_symMatches = elapsed_sortMatches = 0; elapsed_bestMatches = double(best - begin) / CLOCKS_PER_SEC;
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
void MatcherGPU::setGPUMatcher(int _matcher) {                                    // :16-42
    matcherType = _matcher;
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
    if ((_grayImage.cols & 15) == 0 && (_grayImage.rows & 15) == 0) {
        uint8_t* lv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int i = 1; i < 5; i++) { currentFrame->grayImage[i].create(_grayImage.rows >> i, _grayImage.cols >> i, CV_8U); lv[i] = currentFrame->grayImage[i].data; }
        int rc = vis_camera_update(VisDevice::get(), _grayImage.data, _grayImage.cols, _grayImage.rows, (int)_grayImage.step, lv);
        if (rc) VisDevice::fail(rc, "Camera::Update");
    }
}
void Camera::computeGradient() {                                                  // src/Camera.cpp:167-184
    const Mat& img = currentFrame->grayImage[0];
    if ((img.cols & 15) || (img.rows & 15)) return;                               // same restriction as Update's half pyramid
    int16_t* gx[5]; int16_t* gy[5]; uint8_t* g[5];
    for (int lvl = 0; lvl < 5; lvl++) {
        currentFrame->gradientX[lvl].create(img.rows >> lvl, img.cols >> lvl, CV_16S);
        currentFrame->gradientY[lvl].create(img.rows >> lvl, img.cols >> lvl, CV_16S);
        currentFrame->gradient[lvl].create(img.rows >> lvl, img.cols >> lvl, CV_8U);
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
void Camera::saveFrame() { currentFrame->isKeyFrame = true; frameList.push_back(currentFrame); }   // :188-193
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
void CameraGPU::setGPUMatcher(int _matcher) { matcherGPU.setGPUMatcher(_matcher); matcherGPU.setImageDimensions(w_size[0], h_size[0]); }   // :119-123
int CameraGPU::detectAndComputeGPUFeatures() {                                    // :71-117
    keypointsGPU.release(); descriptorsGPU.release(); frameGPU.release();
    vis_ctx* ctx = VisDevice::get();
    const Mat& img = currentFrame->grayImage[0];
    const int cap = 4096;
    vector<vis_keypoint> kps(cap);
    currentFrame->descriptors.create(cap, 32, CV_8U);
    int n = 0;
    const int slot = nextSlot; nextSlot = (nextSlot + 1) % 32;
    int rc = vis_orb_detect_compute(ctx, img.data, img.cols, img.rows, (int)img.step, slot, kps.data(), currentFrame->descriptors.data, cap, &n);
    if (rc) VisDevice::fail(rc, "detectAndComputeGPUFeatures");
    currentFrame->gpuSlot = slot; frameGPU.slot = keypointsGPU.slot = descriptorsGPU.slot = slot;
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
    matcherGPU.slot1 = last->gpuSlot; matcherGPU.slot2 = currentFrame->gpuSlot;
    matcherGPU.computeGPUMatches();
    matcherGPU.computeBestMatches(n_cells);
    matcherGPU.getGoodMatches(last->nextGoodMatches, currentFrame->prevGoodMatches);
    currentFrame->obtainedGoodMatches = true;
}
bool CameraGPU::addGPUKeyframe() {                                                // :138-202
    clock_t cbegin = clock();
    nPointsDetect = detectAndComputeGPUFeatures();
    clock_t cdetect = clock();
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
        elapsed_detect = double(cdetect - cbegin) / CLOCKS_PER_SEC;
        elapsed_computeGoodMatches = double(cgood - cdetect) / CLOCKS_PER_SEC;
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
    if (!raw && vis_pgm_info(f.c_str(), &w, &h) != 0) return Mat();               // imread returns an empty Mat on failure
    if (w < 1 || h < 1) return Mat();
    Mat m(h, w, CV_8U);
    if (vis_image_read(f.c_str(), m.data, (int)m.step, w, h) != 0) return Mat();
    return m;
}
size_t ImageReader::getSize() { return file_names.size(); }
void ImageReader::computeTimeStep() { TimeStep = (double)getImageTime(1) - (double)getImageTime(0); }   // :107-112

// ---------------------------------------------------------------- vi::VISystemGPU (src/VISystemGPU.cpp, src/VISystem.cpp)
namespace vi {
VISystemGPU::VISystemGPU() {}
VISystemGPU::VISystemGPU(int, char*[]) {}                                         // ros::init dropped (out of scope)
VISystemGPU::~VISystemGPU() { cout << "SLAM System shutdown ..." << endl; }
void VISystemGPU::InitializeSystemGPU(double _fx, double _fy, double _cx, double _cy, int w, int h, int num_cells, int detector, int matcher, Mat image) {
    currentImage = image;
    fx = (float)_fx; fy = (float)_fy; cx = (float)_cx; cy = (float)_cy;           // src/VISystemGPU.cpp:57-60
    InitializeCameraGPU(detector, matcher, w, h, num_cells, 3);
    vis_ctx* ctx = VisDevice::get();
    vis_params p; vis_get_params(ctx, &p);
    p.fx = fx; p.fy = fx; p.cx = cx; p.cy = cy;                                   // findEssentialMat(focal = fx): src/VISystem.cpp:1679
    int rc = vis_set_params(ctx, &p);
    if (rc) VisDevice::fail(rc, "InitializeSystemGPU");
    initialized = true;
    cout << "Initializing system ... done" << endl << endl;
}
void VISystemGPU::InitializeCameraGPU(int d, int m, int w, int h, int c, int l) { cameraGPU.initializateCameraGPU(d, m, w, h, c, l); }   // :132-135
void VISystemGPU::AddFrameGPU(Mat _currentImage) {                                // :137-175
    prevImage = currentImage;
    currentImage = _currentImage.clone();
    cameraGPU.Update(_currentImage);
    cameraGPU.addGPUKeyframe();
    num_keyframes = (int)cameraGPU.frameList.size();
    if (cameraGPU.frameList.size() > 1) {
        if (num_keyframes > num_max_keyframes) FreeLastFrameGPU();
        float R[9], t[3];
        EstimatePoseFeaturesRansac(cameraGPU.frameList[cameraGPU.frameList.size() - 2], cameraGPU.frameList[cameraGPU.frameList.size() - 1], R, t);
    }
}
void VISystemGPU::FreeLastFrameGPU() {                                            // :178-182 (without the reference's leak)
    delete cameraGPU.frameList[0];
    cameraGPU.frameList.erase(cameraGPU.frameList.begin());
}
int VISystemGPU::EstimatePoseFeaturesRansac(Frame* prev, Frame* cur, float R_out[9], float t_out[3]) {   // src/VISystem.cpp:1655-1708
    vis_ctx* ctx = VisDevice::get();
    const int m = (int)prev->nextGoodMatches.size();
    vector<float> p1(2 * (size_t)std::max(m, 1)), p2(2 * (size_t)std::max(m, 1));
    for (int i = 0; i < m; i++) {                                                 // KeyPoint::convert, :1673-1674
        p1[2 * i] = prev->nextGoodMatches[i].pt.x; p1[2 * i + 1] = prev->nextGoodMatches[i].pt.y;
        p2[2 * i] = cur->prevGoodMatches[i].pt.x; p2[2 * i + 1] = cur->prevGoodMatches[i].pt.y;
    }
    double E[9], R[9], t[3]; int ninl = 0, iters = 0, ngood = 0;
    int rc = vis_essential_ransac(ctx, p1.data(), p2.data(), m, E, nullptr, &ninl, &iters);   // :1679-1680
    if (rc) VisDevice::fail(rc, "findEssentialMat");
    for (int i = 0; i < 9; i++) R_out[i] = (i % 4 == 0) ? 1.f : 0.f;
    t_out[0] = t_out[1] = t_out[2] = 0.f;
    lastInliers = ninl; lastPoseGood = 0;
    if (ninl > 0) {
        rc = vis_recover_pose(ctx, E, p1.data(), p2.data(), m, R, t, &ngood);      // :1701
        if (rc) VisDevice::fail(rc, "recoverPose");
        for (int i = 0; i < 9; i++) R_out[i] = (float)R[i];                       // convertTo(CV_32FC1), :1702-1703
        for (int i = 0; i < 3; i++) t_out[i] = (float)t[i];
        lastPoseGood = ngood;
    }
    return ninl;
}
}  // namespace vi
