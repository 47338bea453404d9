/*
 * vis_oracle.h -- CPU ORACLE for the vi-slam Camera/Matcher/RANSAC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle (oracle/_build/libvis_oracle.so).  The HIP library never links, includes
 * or calls anything in this directory.
 *
 * What it restates: the arithmetic the reference's CPU path performs THROUGH OpenCV 3.2
 * (cv::ORB::detectAndCompute at /root/reference/src/Camera.cpp:87, BFMatcher::knnMatch
 * at src/Matcher.cpp:86,88, findEssentialMat/recoverPose at src/VISystem.cpp:1680,1701)
 * plus the reference's own post-filters (src/Matcher.cpp:96-367) and F2FRansac
 * (src/VISystem.cpp:612-769).  OpenCV 3.2 is an un-vendored dependency (README.md:3,15;
 * opencv_install.md:47-63) that is absent from this environment, and the reference
 * ships no tests, fixtures or golden vectors:
 *
 *      ****  PARITY UNPINNED versus real OpenCV  ****
 *
 * The OpenCV-owned steps follow the published algorithms as written up in SURVEY.md
 * Appendix A; every function cites the reference call site it stands in for.  The
 * reference cannot be compiled here (needs OpenCV + ROS + Eigen + Ceres), so there is
 * no oracle/_ref build.
 */
#ifndef VIS_ORACLE_H_
#define VIS_ORACLE_H_
#include "../include/vislam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ORB_Impl level sizes / scales / per-level quotas (Appendix A.1 items 2,3) */
int orc_level_geometry(const vis_params* p, int w, int h, int32_t* widths, int32_t* heights,
                       float* scales, int32_t* quotas);
/* cv::resize(..., INTER_LINEAR) 8-bit fixed-point path (Appendix A.1 item 2) */
int orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                      uint8_t* dst, int dw, int dh, int dstride);
/* Camera::Update half pyramid, src/Camera.cpp:68-70 (resize 0.5,0.5 == 2x2 box mean); level sizes = cvRound(size * 0.5) */
void orc_half_pyramid_dims(int w, int h, int32_t lw[5], int32_t lh[5]);
int orc_half_pyramid(const uint8_t* img, int w, int h, int stride, uint8_t* const out_levels[5]);
/* cv::FAST(img, thr, nonmax=true): keypoints in row-major order; also the raw score map
 * (0 for non-corners) when score_map != NULL (w*h bytes). returns count or <0 */
int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold,
                    int32_t* xs, int32_t* ys, int32_t* scores, int cap, uint8_t* score_map);
/* GaussianBlur 7x7 sigma 2, BORDER_REFLECT_101, 8-bit fixed point (Appendix A.1 item 8) */
int orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* cv::ORB::detectAndCompute(img, noArray(), kps, desc), src/Camera.cpp:84-92.
 * Output in canonical order (octave asc, response desc, y asc, x asc). */
int orc_orb_detect_compute(const vis_params* p, const uint8_t* img, int w, int h, int stride,
                           vis_keypoint* kps, uint8_t* desc, int cap, int* n_out);
/* BFMatcher(NORM_HAMMING).knnMatch(k=2) both directions, src/Matcher.cpp:83-94 */
int orc_knn2_hamming(const uint8_t* d1, int n1, const uint8_t* d2, int n2,
                     vis_dmatch* out12, vis_dmatch* out21);
/* Matcher::computeBestMatches, src/Matcher.cpp:353-367 (-> 96-144, 329-352, 171-244) */
int orc_good_matches(const vis_params* p, const vis_keypoint* kps1, int n1,
                     const vis_keypoint* kps2, int n2,
                     const vis_dmatch* knn12, const vis_dmatch* knn21,
                     vis_dmatch* good, int cap, int* n_good,
                     vis_dmatch* sym_out, int sym_cap, int* n_sym);
/* 5-point minimal solver on 5 normalised correspondences: up to 10 E (row-major, unit
 * Frobenius norm, x2^T E x1 = 0), ordered by ascending root z. returns count */
int orc_five_point(const double* q1xy, const double* q2xy, double* Es);
/* test hook: the same call + the degree-10 polynomial det B(z) (11 ascending coefficients) and the real roots the
 * derivative-interlacing bisection found for it (ascending) */
int orc_five_point_poly(const double* q1xy, const double* q2xy, double* Es, double* poly11, double* roots10, int* n_roots);
/* findEssentialMat(..., RANSAC, prob, thr), src/VISystem.cpp:1679-1680 */
int orc_essential_ransac(const vis_params* p, const float* p1xy, const float* p2xy, int m,
                         double E[9], uint8_t* mask, int* n_inliers, int* iters_run);
/* recoverPose, src/VISystem.cpp:1701 */
int orc_recover_pose(const vis_params* p, const double E[9], const float* p1xy,
                     const float* p2xy, int m, double R[9], double t[3], int* n_good);
/* VISystem::F2FRansac, src/VISystem.cpp:612-769 */
int orc_f2f_ransac(const vis_params* p, const vis_keypoint* pts1, const vis_keypoint* pts2, int m,
                   const float rot[9], const int32_t* sample_idx, int iters,
                   float scale, float out_t[3], int* count_max);
/* the sample sequence cv::RNG((uint64)-1) + getSubset would draw: idx5[iters*5] */
int orc_ransac_samples(uint64_t seed, int count, int iters, int32_t* idx5);

/* Camera::computeGradient on ONE level, src/Camera.cpp:171-181: Scharr dx/dy (CV_16S, scale as the reference
 * passes it: 3), |.| saturated to u8, 0.5/0.5 blend rounded half to even.  Outputs dense w x h; any may be NULL */
int orc_scharr_gradient(const uint8_t* img, int w, int h, int stride, int scale,
                        int16_t* gx, int16_t* gy, uint8_t* g);
/* Camera::ObtainPatchesPointsPreviousFrame, src/Camera.cpp:358-410: candidate pixel list of one level
 * (rows x,y,1,1); lw/lh = the 5 half-pyramid level sizes.  n_out = full count even when cap is smaller */
int orc_patch_points(const vis_keypoint* good, int n, const int32_t* lw, const int32_t* lh, int level,
                     float* xyzw, int cap, int* n_out);
/* Camera::ObtainDebugPointsPreviousFrame, src/Camera.cpp:413-445 */
int orc_debug_points(const vis_keypoint* good, int n, int level, float* xyzw, int cap, int* n_out);

/* VISystem::EstimatePoseFeatures, src/VISystem.cpp:1113-1448 (+ WarpFunctionSE3 :1495-1558, InitializePyramid
 * :1451-1493): same arguments as vis_estimate_pose_features; see align.cpp for what is restated and what is SPEC */
void orc_default_align_params(vis_align_params* ap);
int orc_estimate_pose_features(const vis_align_params* ap, int w, int h,
                               const uint8_t* const gray1[5], const uint8_t* const gray2[5],
                               const int16_t* const gx1[5], const int16_t* const gy1[5],
                               const float* const cand1[5], const int32_t n_cand[5],
                               const vis_se3f* init, vis_align_result* out);
/* Sophus::SE3f pieces (thirdparty/sophus/se3.hpp:723-744, :317-321, :253-259), Mat::inv on 6x6 float (LU),
 * VISystem::TukeyFunctionWeights (:1797-1825, not on the live path) */
void orc_se3_exp(const float a[6], vis_se3f* out);
void orc_se3_mul(const vis_se3f* a, const vis_se3f* b, vis_se3f* out);
void orc_se3_from_rt(const float R[9], const float t[3], vis_se3f* out);
void orc_se3_matrix(const vis_se3f* a, float M[16]);
int orc_lu_invert6(const float A[36], float inv[36]);
int orc_tukey_weights(const float* residuals, int n, float* w);

/* unit-test hooks */
void orc_sincos_det(double x, double* s, double* c);
float orc_fast_atan2(float y, float x);
void orc_umax(int half_patch, int* out);
void orc_gaussian_kernel7_q8(int* k);

/* whole per-frame CPU pipeline for the cpu_baseline leg: detect cur, match against prev
 * (knn both directions as the reference does), filters, essential RANSAC, recoverPose. */
typedef struct orc_frame_result {
    int n_kp, n_sym, n_good, n_inliers, n_pose_good, iters_run;
    double E[9], R[9], t[3];
} orc_frame_result;
int orc_pipeline_frame(const vis_params* p, const uint8_t* img, int w, int h, int stride,
                       const vis_keypoint* prev_kps, const uint8_t* prev_desc, int n_prev,
                       vis_keypoint* kps, uint8_t* desc, int cap, orc_frame_result* res);

/* the same pipeline over n resident frames, frame-parallel on `threads` host threads (the 'all cores' CPU baseline) */
int orc_pipeline_stream_mt(const vis_params* p, const uint8_t* frames, int n, int w, int h, int stride,
                           int threads, double* seconds, orc_frame_result* results);

#ifdef __cplusplus
}
#endif
#endif
