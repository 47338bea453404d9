/*
 * vislam_hip.h -- C ABI of the MI355X-native visual front-end (libvislam_hip.so).
 *
 * This is the drop-in boundary for the Camera/Matcher/RANSAC hot path of
 * MecatronicaUSB/vi-slam.  The reference has no FFI of its own: its "GPU path" is
 * three C++ classes that call OpenCV's CUDA module.  Every entry point below
 * replaces one of those OpenCV(-CUDA) call sites; the reference file:line it
 * replaces is cited next to it.  The C++ adapter classes in vi-slam_amd/host/ keep
 * the reference's CameraGPU / MatcherGPU / VISystemGPU surface on top of this ABI.
 *
 * Conventions
 *   - plain C, no torch / OpenCV types; pointers + sizes only.
 *   - return value: VIS_OK (0) or a negative VIS_E_* code; never throws, never exits
 *     (reference convention is cout+exit / cv::Exception; adapters decide).
 *   - a vis_ctx is bound to ONE device and is NOT thread-safe (the reference is
 *     single-threaded, one frame in flight: src/main_vi_slamGPU.cpp:118-123).
 *   - "host" pointers are ordinary CPU memory; "dev" pointers are HIP device
 *     memory on the context's device (e.g. torch tensor .data_ptr()).
 *   - all functions are synchronous w.r.t. their host outputs unless they say
 *     "async": those enqueue on the context stream (vis_set_stream) and return.
 */
#ifndef VISLAM_HIP_H_
#define VISLAM_HIP_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIS_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------- */
enum {
    VIS_OK = 0,
    VIS_E_INVALID = -1,     /* bad argument / shape                            */
    VIS_E_NODEVICE = -2,    /* no HIP device (reference: main_vi_slamGPU.cpp:44-48 returns -1) */
    VIS_E_HIP = -3,         /* a HIP runtime call failed (see vis_last_error)  */
    VIS_E_CAPACITY = -4,    /* caller buffer too small / internal cap exceeded */
    VIS_E_STATE = -5,       /* call order wrong (e.g. slot empty, no plan)     */
    VIS_E_NOMEM = -6
};

/* ---- POD layouts crossing the ABI --------------------------------------- */
/* same field order and size (28 B) as cv::KeyPoint (Frame::keypoints, include/Camera.hpp:50) */
typedef struct vis_keypoint {
    float x, y;          /* level-0 pixel coordinates                            */
    float size;          /* 31 * scale(octave)                                   */
    float angle;         /* degrees [0,360)                                      */
    float response;      /* Harris response                                      */
    int32_t octave;      /* pyramid level                                        */
    int32_t class_id;    /* -1                                                   */
} vis_keypoint;

/* same field order and size (16 B) as cv::DMatch (Matcher::matches, include/Matcher.hpp:52) */
typedef struct vis_dmatch {
    int32_t queryIdx, trainIdx, imgIdx;
    float distance;      /* Hamming distance as float, like BFMatcher           */
} vis_dmatch;

enum { VIS_SYM_REFERENCE_EFFECTIVE = 0,  /* mutual best + ratio on direction 1 only:
                                            what src/Matcher.cpp:96-144 effectively does */
       VIS_SYM_INTENDED = 1 };           /* ratio on both directions + mutual best      */

enum { VIS_DESC_BYTES = 32, VIS_MAX_LEVELS = 16, VIS_MAX_GRID_ROOT = 32 };

/* One POD with every knob of the path.  Defaults (vis_default_params) are the
 * reference's hard-coded constants; see SURVEY.md section 5 "Config / flags". */
typedef struct vis_params {
    /* ORB -- cv::ORB::create(nfeatures) defaults; src/Camera.cpp:127 (200), src/CameraGPU.cpp:99 (1000) */
    int32_t nfeatures;        /* 1000 */
    int32_t nlevels;          /* 8    */
    float   scale_factor;     /* 1.2f */
    int32_t edge_threshold;   /* 31   */
    int32_t patch_size;       /* 31   */
    int32_t fast_threshold;   /* 20   */
    /* matcher post-filters -- src/Matcher.cpp:103 (0.8f), calibration/calibrationEUROC.xml:54 (49) */
    float   ratio;            /* 0.8f */
    int32_t n_cells;          /* 49   */
    int32_t w_size, h_size;   /* Matcher::setImageDimensions, src/Matcher.cpp:30-34 */
    int32_t sym_mode;         /* VIS_SYM_* */
    /* essential RANSAC -- src/VISystem.cpp:1680 (prob 0.999, thr 1.0); OpenCV 3.2 maxIters = 1000 */
    double  ransac_prob;      /* 0.999 */
    double  ransac_threshold; /* 1.0 px */
    int32_t ransac_max_iters; /* 1000 */
    int32_t ransac_adaptive;  /* 1 = OpenCV adaptive stop, 0 = always max_iters (timing runs) */
    uint64_t ransac_seed;     /* 0xFFFFFFFFFFFFFFFF = cv::RNG((uint64)-1) */
    /* intrinsics -- calibration/calibrationEUROC.xml:20 */
    double  fx, fy, cx, cy;
    /* F2FRansac -- src/VISystem.cpp:709 (1000 iterations), :523 (threshold 370) */
    int32_t f2f_iters;        /* 1000 */
    double  f2f_threshold;    /* 370  */
    /* which correspondences the batched pose stage consumes: the reference pipeline feeds the grid-filtered good
     * matches (Frame::next/prevGoodMatches, src/VISystem.cpp:1673-1674); BASELINE config 3 asks for RANSAC on the
     * un-gridded symmetric matches (M up to N) */
    int32_t pose_input;       /* VIS_POSE_GOOD */
    /* Device capacity for the keypoints of ONE frame; 0 = the default, sum over the levels of quota + quota/8 + 32.
     * KeyPointsFilter::retainBest keeps EVERY keypoint tied at its cut, so an image of identical corners (a calibration
     * checkerboard) yields more than nfeatures keypoints.  Beyond the capacity a call returns VIS_E_CAPACITY (never a silent cut);
     * the single-frame entry vis_orb_detect_compute grows the capacity itself up to the `cap` its caller passes. <= 65535. */
    int32_t keypoint_capacity;
} vis_params;
enum { VIS_POSE_GOOD = 0, VIS_POSE_SYM = 1 };

/* wall-clock of the last call's device work, from hipEvents on the context stream
 * (mirrors the elapsed_* members, include/Camera.hpp:121-126, include/Matcher.hpp:63-66) */
typedef struct vis_timings {
    float ms_total;
    float ms_pyramid;      /* resize chain                     */
    float ms_fast;         /* FAST score + NMS, all levels     */
    float ms_select;       /* histogram cut + Harris + top-N   */
    float ms_describe;     /* IC angle + blur + rBRIEF         */
    float ms_knn;          /* both knn directions              */
    float ms_filter;       /* ratio/sym/sort/grid              */
    float ms_pose;         /* essential RANSAC + recoverPose   */
    int32_t launches_fast; /* number of FAST kernel launches in the last call */
    int32_t launches_total;
    float ms_update;       /* Camera::Update half pyramid (VIS_STAGE_UPDATE), 0 when the stage did not run */
    float reserved_;
} vis_timings;

typedef struct vis_ctx vis_ctx;

/* ---- lifetime ------------------------------------------------------------ */
const char* vis_version(void);
const char* vis_strerror(int code);
/* replaces cv::cuda::getCudaEnabledDeviceCount(), src/main_vi_slamGPU.cpp:41 */
int  vis_device_count(void);
/* PCI address "dddd:bb:dd.f" of HIP device `device` (len >= 16): the multi-GPU launchers gather it per rank (generalises the device
 * selection of src/main_vi_slamGPU.cpp:41-43: N ranks must sit on N different devices) */
int  vis_device_pci_bus_id(int device, char* out, int len);
/* replaces cv::cuda::setDevice(0), src/main_vi_slamGPU.cpp:43, plus object construction */
int  vis_create(int device, vis_ctx** out);
void vis_destroy(vis_ctx* ctx);
const char* vis_last_error(vis_ctx* ctx);
void vis_default_params(vis_params* p);
/* replaces cuda::ORB::create(1000) (src/CameraGPU.cpp:99), createBFMatcher(NORM_HAMMING)
 * (src/MatcherGPU.cpp:30), Matcher::setImageDimensions (src/Matcher.cpp:30-34) */
int  vis_set_params(vis_ctx* ctx, const vis_params* p);
int  vis_get_params(vis_ctx* ctx, vis_params* p);
/* enqueue on this hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = context's own stream */
int  vis_set_stream(vis_ctx* ctx, void* hip_stream);
int  vis_last_timings(vis_ctx* ctx, vis_timings* t);
/* per-level geometry the context derived from params for a w x h input
 * (ORB_Impl::detectAndCompute level sizes / quotas) */
int  vis_level_geometry(vis_ctx* ctx, int w, int h, int32_t* widths, int32_t* heights,
                        float* scales, int32_t* quotas);

/* ---- single-frame API: one call per OpenCV(-CUDA) call site --------------- */
/* Sizes of Camera::Update's levels (src/Camera.cpp:68-70: resize(prev, next, Size(), 0.5, 0.5)): cv::resize takes
 * dsize = cvRound(size * 0.5) -- round half to EVEN: 135 -> 68, 137 -> 68 -- so a level can be one row / column larger than the
 * reference's own bookkeeping `w_size[0] >> lvl` (src/Camera.cpp:42-47; 1080 -> 540, 270, 135, 68 against 67).  Both exist
 * here as they do there: buffers, strides, the gradients and the alignment's test of a WARPED point (src/VISystem.cpp:1299:
 * `y2 < image2.rows && x2 < image2.cols`) follow these sizes; the patch builders bound the candidate points they emit by `>> lvl`
 * like the reference.  For sizes that halve exactly four times (752x480) they coincide. */
void vis_half_pyramid_dims(int w, int h, int32_t lw[5], int32_t lh[5]);
/* Camera::Update, src/Camera.cpp:63-72: copy + 4x half-resolution levels.
 * out_levels[l] (l=1..4) receives lw[l]*lh[l] bytes (vis_half_pyramid_dims), tightly packed; out_levels[0] may be NULL.
 * With scale exactly 2 cv::resize(INTER_LINEAR) runs its area-fast path: (a+b+c+d+2)>>2 over every complete 2x2 block; where a
 * level is one larger than half of an odd source size, the last column / row averages the pixels that exist
 * (saturate_cast<uchar>((float)sum / count), round half to even).  w, h >= 16. */
int  vis_camera_update(vis_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                       uint8_t* const out_levels[5]);
/* replaces frameGPU.upload + cuda::ORB::detectAndCompute + descriptorsGPU.download,
 * src/CameraGPU.cpp:81,99-103 (CPU twin: src/Camera.cpp:87).  Keypoints/descriptors stay
 * resident in device slot `frame_slot` for the matcher (Frame list, include/Camera.hpp:104). */
int  vis_orb_detect_compute(vis_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                            int frame_slot, vis_keypoint* kps_out, uint8_t* desc_out,
                            int cap, int* n_out);
/* replaces descriptorsGPU[0/1].upload + 2x cuda knnMatch(k=2), src/MatcherGPU.cpp:49-56
 * (CPU twin src/Matcher.cpp:86,88).  out12: n_q x 2, out21: n_t x 2 (missing neighbours:
 * trainIdx = -1).  n_q/n_t are the slots' keypoint counts. */
int  vis_bf_knn2_hamming(vis_ctx* ctx, int slot_q, int slot_t,
                         vis_dmatch* out12, vis_dmatch* out21);
/* same, on caller-provided host descriptor arrays (n x 32 bytes) */
int  vis_bf_knn2_hamming_host(vis_ctx* ctx, const uint8_t* desc_q, int n_q,
                              const uint8_t* desc_t, int n_t,
                              vis_dmatch* out12, vis_dmatch* out21);
/* fused Matcher::computeBestMatches (ratio, symmetry, y-sort, grid-cell best),
 * src/Matcher.cpp:353-367 -> 96-244; writes goodMatches (<= root^2).  Also returns the
 * symmetric matches when sym_out != NULL (Matcher::matches, capacity sym_cap). */
int  vis_good_matches(vis_ctx* ctx, int slot_prev, int slot_cur,
                      vis_dmatch* good, int cap, int* n_good,
                      vis_dmatch* sym_out, int sym_cap, int* n_sym);
/* same filter chain on host-provided knn results + keypoints (unit-testable piece) */
int  vis_good_matches_host(vis_ctx* ctx, const vis_keypoint* kps1, int n1,
                           const vis_keypoint* kps2, int n2,
                           const vis_dmatch* knn12, const vis_dmatch* knn21,
                           vis_dmatch* good, int cap, int* n_good,
                           vis_dmatch* sym_out, int sym_cap, int* n_sym);
/* replaces cv::findEssentialMat(p1,p2,focal,pp,RANSAC,0.999,1.0), src/VISystem.cpp:1679-1680.
 * p1xy/p2xy: m x 2 floats (pixels).  E row-major. mask may be NULL. */
int  vis_essential_ransac(vis_ctx* ctx, const float* p1xy, const float* p2xy, int m,
                          double E[9], uint8_t* mask, int* n_inliers, int* iters_run);
/* replaces cv::recoverPose(E,p1,p2,R,t,focal,pp), src/VISystem.cpp:1701 */
int  vis_recover_pose(vis_ctx* ctx, const double E[9], const float* p1xy, const float* p2xy,
                      int m, double R[9], double t[3], int* n_good);
/* VISystem::F2FRansac, src/VISystem.cpp:612-769.  rot: 3x3 row-major f32 (IMU rotation),
 * sample_idx: 2*iters explicit sample indices (the reference uses unseeded rand(), :712-713),
 * scale: |t_GT|.  out: 3 floats. */
int  vis_f2f_ransac(vis_ctx* ctx, const vis_keypoint* pts1, const vis_keypoint* pts2, int m,
                    const float rot[9], const int32_t* sample_idx, int iters,
                    float scale, float out_t[3], int* count_max);

/* ---- the step after matching in CameraGPU::addGPUKeyframe (src/CameraGPU.cpp:154-157) ---- */
/* Camera::Update's half pyramid (src/Camera.cpp:63-72) + Camera::computeGradient (src/Camera.cpp:167-184) for n
 * frames resident in HBM.  Per frame and per level l = 0..4 of lw[l] x lh[l] pixels (vis_half_pyramid_dims): Scharr dx and dy as
 * CV_16S with OpenCV's `scale` argument (the reference call Scharr(img, g, CV_16S, 1, 0, 3, 0, BORDER_DEFAULT)
 * passes scale = 3, delta = 0), and gradient = addWeighted(|dx| sat u8, 0.5, |dy| sat u8, 0.5, 0).
 * All outputs are caller-owned DEVICE buffers of n * vis_gradient_frame_elems(w, h) elements: inside a frame the
 * levels are dense and back to back (level l starts at sum_{k<l} lw[k] lh[k]); d_gray receives levels 1..4 of
 * the half pyramid (its level-0 part is left untouched: level 0 is the frame itself).  w, h >= 16,
 * stride % 4 == 0, 1 <= scale <= 8 (int16 cannot overflow), buffers 16-byte aligned. */
size_t vis_gradient_frame_elems(int w, int h);
int  vis_gradient_batch(vis_ctx* ctx, const uint8_t* d_frames, int w, int h, int stride, int n, int scale,
                        uint8_t* d_gray, int16_t* d_gx, int16_t* d_gy, uint8_t* d_g);
/* one host frame; out pointers per level may be NULL; each receives (w>>l)*(h>>l) elements */
int  vis_compute_gradient(vis_ctx* ctx, const uint8_t* img, int w, int h, int stride, int scale,
                          int16_t* const gx[5], int16_t* const gy[5], uint8_t* const g[5]);
/* Camera::ObtainPatchesPointsPreviousFrame (src/Camera.cpp:358-410) and ObtainDebugPointsPreviousFrame
 * (:413-445): per level l the candidate list as rows (x, y, 1, 1) in the reference's push_back order, built from
 * at most 200 matched keypoints of the previous keyframe.  patch[l] / debug[l] receive up to `cap` rows
 * (VIS_E_CAPACITY if a level has more; n_patch[l] then holds the required count).  Level sizes come from
 * params.w_size >> l, params.h_size >> l (CameraModel w_size[lvl], h_size[lvl]). */
int  vis_patch_points(vis_ctx* ctx, const vis_keypoint* good, int n, int cap,
                      float* const patch[5], int n_patch[5], float* const debug[5], int n_debug[5]);

/* ---- the pose step the GPU main calls: VISystem::EstimatePoseFeatures (src/VISystem.cpp:1113-1448), called from
 * VISystemGPU::AddFrameGPU (src/VISystemGPU.cpp:167) -- SURVEY 8(f) N4 --------------------------------------- */
/* Gauss-Newton photometric alignment of the candidate points of the previous keyframe (ObtainPatchesPoints...) to the
 * current frame, coarse to fine over the half pyramid; pose = Sophus::SE3f (unit quaternion + translation). */
typedef struct vis_se3f { float qx, qy, qz, qw; float tx, ty, tz; } vis_se3f;   /* Sophus::SE3f storage order */
typedef struct vis_align_params {
    float fx, fy, cx, cy;        /* level-0 intrinsics, the float members of VISystem (include/VISystem.hpp:82) */
    int32_t first_level;         /* 3      src/VISystem.cpp:1119 */
    int32_t last_level;          /* 0      :1120 */
    int32_t max_iterations;      /* 10     :1117 */
    float   epsilon;             /* 0.001f :1115 */
    float   z_factor;            /* 0.002f :1121 */
} vis_align_params;
typedef struct vis_align_result {
    vis_se3f pose;               /* current_pose after the last level (Frame::rigid_transformation_, :1445) */
    float matrix[16];            /* pose.matrix(), row-major 4x4 */
    float error[5];              /* last mean squared residual per level */
    float initial_error;
    int32_t iterations[5];       /* iteration index k at which the level stopped */
    int32_t n_residuals[5];      /* valid residuals in the last iteration of the level */
} vis_align_result;
void vis_default_align_params(vis_align_params* ap);
/* Sophus::SE3f value operations on the host (thirdparty/sophus/se3.hpp:723-744 exp, :317-321 product, :253-259 matrix,
 * SE3(Matrix3, Point)): what VISystem::Track / EstimatePoseFeatures compose poses with (src/VISystem.cpp:1413,1607);
 * the same code the alignment kernel runs.  a = (upsilon, omega); matrices row-major. */
void vis_se3_exp(const float a[6], vis_se3f* out);
void vis_se3_mul(const vis_se3f* a, const vis_se3f* b, vis_se3f* out);
void vis_se3_from_rt(const float R[9], const float t[3], vis_se3f* out);
void vis_se3_matrix(const vis_se3f* a, float M[16]);
/* One pair, HOST pointers.  Level l images are dense (w>>l) x (h>>l): gray1/gx1/gy1 of the previous keyframe
 * (Frame::grayImage / gradientX / gradientY), gray2 of the current frame, cand1[l] = n_cand[l] rows (x, y, z, 1) as
 * Frame::candidatePoints[l] holds them.  Levels outside [last_level, first_level] may be NULL.  init may be NULL
 * (identity); the reference seeds it from the IMU rotation residual and the ground-truth translation (:1133-1166). */
int  vis_estimate_pose_features(vis_ctx* ctx, const vis_align_params* ap, int w, int h,
                                const uint8_t* const gray1[5], const uint8_t* const gray2[5],
                                const int16_t* const gx1[5], const int16_t* const gy1[5],
                                const float* const cand1[5], const int32_t n_cand[5],
                                const vis_se3f* init, vis_align_result* out);
/* Batched, DEVICE pointers: n consecutive frames resident in HBM and the outputs of vis_gradient_batch for the same
 * frames (d_gray levels 1..4, d_gx, d_gy).  Pair i = (frame i-1 -> frame i), i = 1..n-1; d_out[0] is zeroed.  The
 * candidate points of pair i are generated on the fly from d_pts: max_pts (x, y) floats per pair = the matched
 * keypoints of frame i-1 (Frame::nextGoodMatches, at most 200 are used like the reference), d_npts[i] of them valid.
 * d_init: n poses or NULL.  Asynchronous on the context's stream.  w, h >= 16. */
int  vis_align_batch(vis_ctx* ctx, const vis_align_params* ap, const uint8_t* d_frames, int w, int h, int stride, int n,
                     const uint8_t* d_gray, const int16_t* d_gx, const int16_t* d_gy,
                     const float* d_pts, const int32_t* d_npts, int max_pts,
                     const vis_se3f* d_init, vis_align_result* d_out);
/* the same on the pairs of the last vis_batch_run (stages must have included MATCH): the matched points come from the
 * plan (the grid-filtered good matches of every pair).  Pair 0 (frame 0 against the carried frame) is skipped.
 * Runs on the context's POSE stream, ordered behind everything queued on the context's stream so far (the gradients) and
 * behind the matcher, so that it overlaps the next vis_batch_run: d_frames, the gradient buffers and d_out are in use until
 * vis_batch_sync -- or until a later vis_gradient_batch / vis_batch_align / vis_feeder_submit of this context, which wait for it. */
int  vis_batch_align(vis_ctx* ctx, const vis_align_params* ap, const uint8_t* d_frames, int n,
                     const uint8_t* d_gray, const int16_t* d_gx, const int16_t* d_gy,
                     const vis_se3f* d_init, vis_align_result* d_out);

/* ---- frame ingest (src/ImageReader.cpp) ------------------------------------- */
/* ImageReader::searchImages (src/ImageReader.cpp:49-74): the .pgm / .raw / .png files of `dir` in byte order, names
 * separated by '\n' in names_out (cap_bytes); *count = number of files.  names_out may be NULL to only count.
 * ("." and ".." are skipped by name; the reference erases the first two sorted entries.) */
int  vis_image_list(const char* dir, char* names_out, int cap_bytes, int* count);
/* ImageReader::getImageTime (:41-47): atol of the file name's stem (EuRoC names its images <timestamp ns>.ext) */
long vis_image_time(const char* file_name);
/* stand-in for imread(..., CV_LOAD_IMAGE_GRAYSCALE) (:80-82) on the formats this build reads: binary PGM
 * (P5, maxval <= 255, '#' comments allowed), headerless raw (w*h bytes) and greyscale PNG (what EuRoC ships: colour
 * type 0 or 4, 8 or 16 bit -- the high byte --, non-interlaced; chunk CRCs checked; colour / interlaced PNGs are
 * refused with VIS_E_INVALID).  vis_image_info tells PGM from PNG by the magic bytes. */
int  vis_pgm_info(const char* path, int* w, int* h);
int  vis_image_info(const char* path, int* w, int* h);
int  vis_image_read(const char* path, uint8_t* out, int out_stride, int w, int h);
/* Pinned-host double-buffered H2D feeder for vis_batch_run: fill vis_feeder_host_buffer(f, k) (batch x h x w,
 * dense; blocks while an earlier copy out of it is in flight), vis_feeder_submit(f, k, n, &d) enqueues the copy on a
 * copy stream and orders the context's detect stream after it, run vis_batch_run(ctx, d, n, ...), then
 * vis_feeder_release(f, k).  Alternating k = 0, 1 overlaps the copy of batch i+1 with the processing of batch i. */
typedef struct vis_feeder vis_feeder;
int  vis_feeder_create(vis_ctx* ctx, int w, int h, int batch, vis_feeder** out);
void vis_feeder_destroy(vis_feeder* f);
uint8_t* vis_feeder_host_buffer(vis_feeder* f, int which);
int  vis_feeder_submit(vis_feeder* f, int which, int n, const uint8_t** d_frames);
int  vis_feeder_release(vis_feeder* f, int which);

/* ---- batched stream API (throughput path) --------------------------------- */
/* Plan device buffers for batches of up to `max_frames` w x h frames.  Frames in a
 * batch are consecutive frames of ONE camera stream: frame i is matched against frame
 * i-1 (Camera::computeGoodMatches: query = last saved frame, src/Camera.cpp:146-157);
 * frame 0 is matched against the last frame of the previous vis_batch_run call
 * (carried on device), or not at all after vis_batch_reset. */
int  vis_batch_plan(vis_ctx* ctx, int w, int h, int stride, int max_frames);
int  vis_batch_reset(vis_ctx* ctx);
enum { VIS_STAGE_DETECT = 1, VIS_STAGE_MATCH = 2, VIS_STAGE_POSE = 4, VIS_STAGE_ALL = 7,
       /* Camera::Update (src/Camera.cpp:63-72) for every frame of the batch: the 4 half-resolution levels into a plan-owned
        * buffer (vis_batch_half_pyramid), on a side stream that starts behind the pyramid launches of the detect chain and is joined
        * by the detect stream at the end of its chain. */
       VIS_STAGE_UPDATE = 8, VIS_STAGE_FRAME = 15,
       /* Camera::computeGradient (src/Camera.cpp:167-184; inside addGPUKeyframe, src/CameraGPU.cpp:154) for every frame of the
        * batch: Scharr dx / dy (CV_16S, scale 3) and the blended magnitude on the 5 half-pyramid levels, into plan-owned buffers
        * (vis_batch_gradients), on the same side stream as Camera::Update -- pure streaming work beside the detect chain.
        * Implies VIS_STAGE_UPDATE.  Overwrites the previous batch's gradients: it waits for a vis_batch_align still reading them. */
       VIS_STAGE_GRADIENT = 16 };
/* Asynchronous: d_frames = n_frames images resident in HBM (dev ptr, row stride from the plan, frame stride =
 * stride*h).  The context runs three streams: the detect chain (the stream set with vis_set_stream / the context's
 * own), the matcher, and the RANSAC/pose stage; consecutive calls overlap (detect of batch i+1 with match and pose of
 * batch i).  d_frames may be reused once the detect chain of this call has finished (vis_batch_sync, or an event
 * recorded on the detect stream after the call; vis_feeder_release does exactly that). */
int  vis_batch_run(vis_ctx* ctx, const uint8_t* d_frames, int n_frames, int stages);
int  vis_batch_sync(vis_ctx* ctx);
/* the half pyramids the last vis_batch_run(... | VIS_STAGE_UPDATE) wrote: DEVICE pointer to n * *frame_elems bytes, laid out
 * like d_gray of vis_gradient_batch (levels dense and back to back inside a frame, level 0's part untouched);
 * VIS_E_STATE if the stage has not run.  Valid until the next vis_batch_run / vis_batch_plan. */
int  vis_batch_half_pyramid(vis_ctx* ctx, const uint8_t** d_half, size_t* frame_elems);
/* the gradients the last vis_batch_run(... | VIS_STAGE_GRADIENT) wrote: DEVICE pointers laid out like the outputs of
 * vis_gradient_batch (d_gray = the half pyramid of vis_batch_half_pyramid); any pointer may be NULL.  VIS_E_STATE if the
 * stage has not run.  Valid until the next vis_batch_run / vis_batch_plan (the plan owns two sets and fills them in turn, so
 * that a vis_batch_align still reading one does not hold up the next step's gradients).  vis_batch_align takes them when its
 * three gradient arguments are NULL. */
int  vis_batch_gradients(vis_ctx* ctx, const uint8_t** d_gray, const int16_t** d_gx, const int16_t** d_gy, const uint8_t** d_g,
                         size_t* frame_elems);
/* copy results of the last batch to host (synchronises). Any pointer may be NULL. */
int  vis_batch_get_keypoints(vis_ctx* ctx, int frame, vis_keypoint* kps, uint8_t* desc,
                             int cap, int* n_out);
int  vis_batch_get_knn(vis_ctx* ctx, int frame, vis_dmatch* out12, int cap12, int* n12,
                       vis_dmatch* out21, int cap21, int* n21);
int  vis_batch_get_matches(vis_ctx* ctx, int frame, vis_dmatch* good, int cap, int* n_good,
                           int* n_sym);
int  vis_batch_get_pose(vis_ctx* ctx, int frame, double E[9], double R[9], double t[3],
                        int* n_inliers, int* n_pose_good, int* iters_run);
/* diagnostics: out[0] = kernel launches of this process so far, out[1] = times a single-frame entry point of this context blocked on the
 * device, out[2] = asynchronous copies those entry points queued, out[3] = 0.  (bench.py `single_frame_api`: per-frame differences.) */
int  vis_debug_counters(vis_ctx* ctx, unsigned long long out[4]);
/* the inlier mask of findEssentialMat (src/VISystem.cpp:1680, the `mask` argument) for pair `frame` of the last batch: one byte per
 * correspondence the pose stage saw, in the order it saw them (good matches, or the symmetric matches with VIS_POSE_SYM).
 * VIS_E_CAPACITY if cap < *n_points (which is still returned). */
int  vis_batch_get_inlier_mask(vis_ctx* ctx, int frame, uint8_t* mask, int cap, int* n_points);
/* what the pose stage leaves per pair on the device (vis_batch_results_async copies these records) */
typedef struct vis_pose_result {
    double E[9], R[9], t[3];
    int32_t n_inliers, n_pose_good, iters_run, n_points;
    int32_t n_models;          /* candidate essential matrices scored against the n_points correspondences (SURVEY 8(d):
                                  point evaluations = n_models x n_points) */
    int32_t reserved_;
} vis_pose_result;
/* Queue the device-to-host copy of the last batch's results -- n pose records, the good matches (n x root^2, dense
 * rows) and their counts -- behind the batch's own work; any pointer may be NULL; pinned host memory makes the copy
 * overlap the next vis_batch_run (pinned = device-accessible, e.g. hipHostMalloc: such destinations are written by one
 * small kernel of the library, which costs the pipeline nothing; anything else goes through hipMemcpyAsync).  n_cap = the number of frames the caller's buffers hold: VIS_E_CAPACITY (nothing is
 * copied) if the last batch had more.  The reference downloads its results every frame (src/CameraGPU.cpp:103, the
 * DMatch vectors of src/MatcherGPU.cpp:54-56).  The copies are only QUEUED: the host may read the buffers after
 * vis_batch_sync() (or after waiting for an event recorded behind this call); a later vis_batch_results_async is
 * ordered behind this one on the device but does not make this one visible to the host by itself. */
int  vis_batch_results_async(vis_ctx* ctx, vis_pose_result* h_pose, vis_dmatch* h_good, int32_t* h_ngood, int n_cap);
/* Batched streams run FAST at a per-level threshold tau >= fast_threshold predicted from the previous batch: retainBest(2 * quota)
 * only keeps corners whose score reaches a cut far above fast_threshold, and a corner below the cut can neither be kept nor
 * suppress a kept one, so any tau <= cut gives the identical keypoints.  The prediction is verified per (frame, level) on the
 * device and whatever it got wrong is redone at fast_threshold inside the same vis_batch_run: results never depend on it.
 * tau_next (nlevels ints, may be NULL) = the thresholds the NEXT batch will start from; *n_redone = (frame, level) pairs the last
 * batch had to redo.  Synchronises.  vis_batch_reset() forgets the prediction. */
int  vis_batch_fast_thresholds(vis_ctx* ctx, int32_t* tau_next, int32_t* n_redone);
/* device-side error/overflow flags of the last batch (0 = clean) */
int  vis_batch_status(vis_ctx* ctx, int* flags);

/* ---- synthetic EuRoC-shaped stream ("S-752") -------- */
/* Integer-only generator, identical bytes on every machine.  canvas: canvas_dim^2 bytes. */
int  vis_synth_canvas(uint8_t* canvas, int canvas_dim, uint64_t seed);
int  vis_synth_frame(const uint8_t* canvas, int canvas_dim, uint64_t seed, int t,
                     int w, int h, uint8_t* out, int out_stride);
/* "S-752P": the same stream with a second depth layer (1.5x parallax, same direction) and independently moving objects
 * (outliers) on top of the static background: image motion that is NOT one planar shift, so the adaptive RANSAC stop
 * does not trigger after a handful of hypotheses.  Same integer-only rule on host and device. */
int  vis_synth_frame_parallax(const uint8_t* canvas, int canvas_dim, uint64_t seed, int t,
                              int w, int h, uint8_t* out, int out_stride);
/* n consecutive frames t0 .. t0+n-1 generated on the device into d_out (frame stride = stride*h); d_canvas = the canvas
 * of vis_synth_canvas in device memory.  mode 0 = S-752, 1 = S-752P.  Byte-identical to the host generators.
 * Asynchronous on the context's stream. */
int  vis_synth_frames_device(vis_ctx* ctx, const uint8_t* d_canvas, int canvas_dim, uint64_t seed, int t0, int n,
                             int w, int h, int stride, int mode, uint8_t* d_out);

#ifdef __cplusplus
}
#endif
#endif /* VISLAM_HIP_H_ */
