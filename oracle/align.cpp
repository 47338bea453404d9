/*
 * align.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, see vis_oracle.h) for the pose step the GPU main actually
 * calls:
 *   VISystem::EstimatePoseFeatures   /root/reference/src/VISystem.cpp:1113-1448   Gauss-Newton photometric alignment
 *   VISystem::WarpFunctionSE3        /root/reference/src/VISystem.cpp:1495-1558
 *   VISystem::InitializePyramid      /root/reference/src/VISystem.cpp:1451-1493   per-level intrinsics
 *   VISystem::IdentityWeights        /root/reference/src/VISystem.cpp:1561-1565   (TukeyFunctionWeights :1797-1825 is
 *                                    commented out at the call site :1328; restated below for completeness)
 * called from VISystemGPU::AddFrameGPU (/root/reference/src/VISystemGPU.cpp:167).
 *
 * PARITY UNPINNED versus the real dependencies: the reference computes this through cv::Mat expression templates
 * (cv::gemm, Mat::inv = LU) and Sophus::SE3f (Eigen quaternions); OpenCV 3.2, Eigen and the std:: sin/cos of the
 * reference's libm are absent here and the reference holds no fixture for this function.  What is restated:
 *   - the control flow, validity tests, Jacobian formulas and constants exactly as written in the reference;
 *   - cv::gemm on CV_32F operands: products and sums in double, one rounding to float at the end [external];
 *   - Mat::inv() (DECOMP_LU) on the 6x6 float normal matrix: Gaussian elimination with partial pivoting in float,
 *     pivot threshold FLT_EPSILON*10, singular -> zero matrix [external: hal::LU32f];
 *   - Sophus::SE3f::exp / operator* / matrix(), SO3f::operator*= renormalisation 2/(1+|q|^2), Eigen's quaternion
 *     product, _transformVector and toRotationMatrix operation order [external], epsilon<float> = 1e-5
 *     (thirdparty/sophus/common.hpp:154-157, so3.hpp:534-568,338-355, se3.hpp:317-321,723-744);
 *   - sin/cos: the oracle's deterministic sincos_det (double, rounded to float) instead of libm sinf/cosf so that
 *     the HIP kernel can reproduce it bit for bit.
 * SPEC decisions where the reference is undefined (SURVEY.md section 0.10 style):
 *   - image2.at<uchar>(round(y2), round(x2)) can index one past the last row/column when y2 > rows - 0.5
 *     (the guard is y2 < rows, :1280): the rounded index is clamped to the last row / column of the Mat;
 *   - zero valid residuals (Residuals.rows == 0 -> 1/0 and an empty gemm, :1331-1333): the level stops;
 *   - the A^T A / A^T r reductions use a FIXED summation order so that a parallel implementation can match:
 *     ALIGN_LANES = 256 partial sums (row i goes to partial i % 256, rows in ascending order); inside each group of
 *     64 partials a binary tree (stride 32, 16, ... 1), then (G0 + G1) + (G2 + G3); all in double (a 256-thread
 *     workgroup = 4 wavefronts of 64 lanes reduces exactly like this).  cv::gemm's own order is sequential; the difference is in the last
 *     bits of a double before the rounding to float.
 */
#include "vis_oracle.h"
#include "oracle_internal.h"
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

const int ALIGN_LANES = 256;

struct Quat { float w, x, y, z; };
struct Se3 { Quat q; float t[3]; };

inline float sin_det(float a) { double s, c; orc::sincos_det((double)a, &s, &c); return (float)s; }
inline float cos_det(float a) { double s, c; orc::sincos_det((double)a, &s, &c); return (float)c; }

// Eigen::Quaternion product
Quat qmul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
// Eigen::QuaternionBase::_transformVector: uv = 2 (q.vec x v); v + w uv + q.vec x uv
void qrot(const Quat& q, const float v[3], float out[3]) {
    float uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const float c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
    out[0] = v[0] + q.w * uv[0] + c[0];
    out[1] = v[1] + q.w * uv[1] + c[1];
    out[2] = v[2] + q.w * uv[2] + c[2];
}
// Eigen::QuaternionBase::toRotationMatrix (row-major R[9])
void qmat(const Quat& q, float R[9]) {
    const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}
// Eigen's rotation-matrix -> quaternion (QuaternionBase::operator=(MatrixBase)), used by SE3(Matrix3, Point)
Quat qfrom(const float R[9]) {
    Quat q;
    float t = R[0] + R[4] + R[8];
    if (t > 0.f) {
        t = std::sqrt(t + 1.f);
        q.w = 0.5f * t;
        t = 0.5f / t;
        q.x = (R[7] - R[5]) * t; q.y = (R[2] - R[6]) * t; q.z = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.f);
        float v[3];
        v[i] = 0.5f * t;
        t = 0.5f / t;
        q.w = (R[3 * k + j] - R[3 * j + k]) * t;
        v[j] = (R[3 * j + i] + R[3 * i + j]) * t;
        v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

// Sophus::SE3f::exp, thirdparty/sophus/se3.hpp:723-744 with SO3::expAndTheta so3.hpp:534-568; a = (upsilon, omega)
Se3 se3_exp(const float a[6]) {
    const float* ups = a; const float* om = a + 3;
    const float theta_sq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2];
    const float theta = std::sqrt(theta_sq);
    const float half = 0.5f * theta;
    float imag, real;
    const float eps = 1e-5f;
    if (theta < eps) {
        const float po4 = theta_sq * theta_sq;
        imag = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * po4;
        real = 1.f - (float)(1.0 / 8.0) * theta_sq + (float)(1.0 / 384.0) * po4;
    } else {
        imag = sin_det(half) / theta;
        real = cos_det(half);
    }
    Se3 r;
    r.q.w = real; r.q.x = imag * om[0]; r.q.y = imag * om[1]; r.q.z = imag * om[2];
    // Omega = hat(omega), Omega^2
    const float O[9] = {0.f, -om[2], om[1], om[2], 0.f, -om[0], -om[1], om[0], 0.f};
    float O2[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    float V[9];
    if (theta < eps) qmat(r.q, V);
    else {
        const float ca = (1.f - cos_det(theta)) / theta_sq;
        const float cb = (theta - sin_det(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + ca * O[i] + cb * O2[i];
    }
    for (int i = 0; i < 3; i++) r.t[i] = V[3 * i] * ups[0] + V[3 * i + 1] * ups[1] + V[3 * i + 2] * ups[2];
    return r;
}
// SE3Base::operator*=, se3.hpp:317-321 + SO3Base::operator*=, so3.hpp:338-355
Se3 se3_mul(const Se3& a, const Se3& b) {
    Se3 r = a;
    float rt[3];
    qrot(a.q, b.t, rt);
    r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
    r.q = qmul(a.q, b.q);
    const float sn = r.q.w * r.q.w + r.q.x * r.q.x + r.q.y * r.q.y + r.q.z * r.q.z;
    if (sn != 1.f) { const float s = 2.f / (1.f + sn); r.q.w *= s; r.q.x *= s; r.q.y *= s; r.q.z *= s; }
    return r;
}

// hal::LU32f as cv::invert(DECOMP_LU) uses it on an n x n float matrix: A is destroyed, B (n x n) starts as identity
bool lu_invert6(float A[36], float B[36]) {
    const int n = 6;
    for (int i = 0; i < 36; i++) B[i] = (i % 7 == 0) ? 1.f : 0.f;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (std::fabs(A[j * n + i]) > std::fabs(A[k * n + i])) k = j;
        if (std::fabs(A[k * n + i]) < FLT_EPSILON * 10) return false;
        if (k != i) {
            for (int j = i; j < n; j++) { const float t = A[i * n + j]; A[i * n + j] = A[k * n + j]; A[k * n + j] = t; }
            for (int j = 0; j < n; j++) { const float t = B[i * n + j]; B[i * n + j] = B[k * n + j]; B[k * n + j] = t; }
        }
        const float d = -1.f / A[i * n + i];
        for (int j = i + 1; j < n; j++) {
            const float alpha = A[j * n + i] * d;
            for (int c = i + 1; c < n; c++) A[j * n + c] += alpha * A[i * n + c];
            for (int c = 0; c < n; c++) B[j * n + c] += alpha * B[i * n + c];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            float s = B[i * n + j];
            for (int k = i + 1; k < n; k++) s -= A[i * n + k] * B[k * n + j];
            B[i * n + j] = s / A[i * n + i];
        }
    return true;
}

struct LevelK { float fx, fy, cx, cy, invfx, invfy; };
// VISystem::InitializePyramid, :1451-1493
void level_intrinsics(const vis_align_params& ap, LevelK K[5]) {
    K[0].fx = ap.fx; K[0].fy = ap.fy; K[0].cx = ap.cx; K[0].cy = ap.cy;
    for (int l = 1; l < 5; l++) {
        K[l].fx = (float)((double)K[l - 1].fx * 0.5);
        K[l].fy = (float)((double)K[l - 1].fy * 0.5);
        K[l].cx = (float)(((double)K[0].cx + 0.5) / (double)(1 << l) - 0.5);
        K[l].cy = (float)(((double)K[0].cy + 0.5) / (double)(1 << l) - 0.5);
    }
    for (int l = 0; l < 5; l++) { K[l].invfx = 1.f / K[l].fx; K[l].invfy = 1.f / K[l].fy; }
}

// 27 sums of one iteration: upper triangle of J^T J (21), J^T r (6); fixed order (see header)
void reduce_normal_equations(const std::vector<float>& J, const std::vector<float>& r, double out[27]) {
    std::vector<double> part((size_t)ALIGN_LANES * 27, 0.0);
    const size_t n = r.size();
    for (size_t i = 0; i < n; i++) {
        double* p = &part[(i % ALIGN_LANES) * 27];
        const float* j = &J[6 * i];
        int s = 0;
        for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) p[s++] += (double)j[a] * (double)j[b];
        for (int a = 0; a < 6; a++) p[21 + a] += (double)j[a] * (double)r[i];
    }
    // inside each group of 64 partials: strides 32, 16, ... 1; then (G0 + G1) + (G2 + G3)
    for (int g = 0; g < ALIGN_LANES; g += 64)
        for (int off = 32; off > 0; off >>= 1)
            for (int t = 0; t < off; t++) for (int s = 0; s < 27; s++) part[(size_t)(g + t) * 27 + s] += part[(size_t)(g + t + off) * 27 + s];
    for (int s = 0; s < 27; s++) out[s] = (part[s] + part[(size_t)64 * 27 + s]) + (part[(size_t)128 * 27 + s] + part[(size_t)192 * 27 + s]);
}

}  // namespace

extern "C" void orc_default_align_params(vis_align_params* ap) {
    if (!ap) return;
    std::memset(ap, 0, sizeof(*ap));
    ap->fx = 458.654f; ap->fy = 457.296f; ap->cx = 367.215f; ap->cy = 248.375f;
    ap->first_level = 3; ap->last_level = 0; ap->max_iterations = 10;      // :1117,1119,1120
    ap->epsilon = 0.001f; ap->z_factor = 0.002f;                             // :1115,1121
}

extern "C" void orc_se3_exp(const float a[6], vis_se3f* out) {
    const Se3 e = se3_exp(a);
    out->qx = e.q.x; out->qy = e.q.y; out->qz = e.q.z; out->qw = e.q.w;
    out->tx = e.t[0]; out->ty = e.t[1]; out->tz = e.t[2];
}
extern "C" void orc_se3_mul(const vis_se3f* a, const vis_se3f* b, vis_se3f* out) {
    const Se3 A = {{a->qw, a->qx, a->qy, a->qz}, {a->tx, a->ty, a->tz}}, B = {{b->qw, b->qx, b->qy, b->qz}, {b->tx, b->ty, b->tz}};
    const Se3 e = se3_mul(A, B);
    out->qx = e.q.x; out->qy = e.q.y; out->qz = e.q.z; out->qw = e.q.w;
    out->tx = e.t[0]; out->ty = e.t[1]; out->tz = e.t[2];
}
extern "C" void orc_se3_from_rt(const float R[9], const float t[3], vis_se3f* out) {
    const Quat q = qfrom(R);
    out->qx = q.x; out->qy = q.y; out->qz = q.z; out->qw = q.w; out->tx = t[0]; out->ty = t[1]; out->tz = t[2];
}
extern "C" void orc_se3_matrix(const vis_se3f* a, float M[16]) {
    const Quat q = {a->qw, a->qx, a->qy, a->qz};
    float R[9]; qmat(q, R);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) M[4 * i + j] = R[3 * i + j]; }
    M[3] = a->tx; M[7] = a->ty; M[11] = a->tz; M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
}
extern "C" int orc_lu_invert6(const float A[36], float inv[36]) {
    float a[36]; std::memcpy(a, A, sizeof(a));
    if (!lu_invert6(a, inv)) { std::memset(inv, 0, 36 * sizeof(float)); return 0; }
    return 1;
}

// VISystem::TukeyFunctionWeights / MedianAbsoluteDeviation / MedianMat, :1797-1870.  Not on the live path (the call at
// :1328 is commented out, IdentityWeights is used); kept as a unit-testable restatement.  MedianMat converts the float
// residuals to CV_8U with saturation (cvRound) before the 256-bin histogram.
static float median_mat_u8(const float* v, int n) {
    int hist[256] = {0};
    for (int i = 0; i < n; i++) { long r = std::lrint((double)v[i]); r = r < 0 ? 0 : (r > 255 ? 255 : r); hist[r]++; }
    const float m = (float)(n / 2);
    int bin = 0;
    for (int i = 0; i < 256; i++) { bin += hist[i]; if ((float)bin > m) return (float)i; }
    return -1.f;
}
extern "C" int orc_tukey_weights(const float* residuals, int n, float* w) {
    if (!residuals || !w || n < 0) return VIS_E_INVALID;
    const float b = 4.6851f;
    const float med = median_mat_u8(residuals, n);
    std::vector<float> dev((size_t)n);
    for (int i = 0; i < n; i++) dev[i] = std::fabs(residuals[i] - med);
    float MAD = 1.4826f * median_mat_u8(dev.data(), n);
    if (MAD == 0) MAD = 1;
    const float inv_MAD = (float)(1.0 / MAD), inv_b2 = (float)(1.0 / (b * b));
    for (int i = 0; i < n; i++) {
        const float x = residuals[i] * inv_MAD;
        if (std::fabs(x) <= b) { const float t = (float)(1.0 - (x * x) * inv_b2); w[i] = t * t; } else w[i] = 0.f;
    }
    return VIS_OK;
}

extern "C" int orc_estimate_pose_features(const vis_align_params* ap, int w, int h,
                                          const uint8_t* const gray1[5], const uint8_t* const gray2[5],
                                          const int16_t* const gx1[5], const int16_t* const gy1[5],
                                          const float* const cand1[5], const int32_t n_cand[5],
                                          const vis_se3f* init, vis_align_result* out) {
    if (!ap || !out || !gray1 || !gray2 || !gx1 || !gy1 || !cand1 || !n_cand) return VIS_E_INVALID;
    if (ap->first_level < ap->last_level || ap->first_level > 4 || ap->last_level < 0 || ap->max_iterations < 1 || w < 16 || h < 16) return VIS_E_INVALID;
    LevelK K[5]; level_intrinsics(*ap, K);
    Se3 pose = {{1.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (init) { pose.q = {init->qw, init->qx, init->qy, init->qz}; pose.t[0] = init->tx; pose.t[1] = init->ty; pose.t[2] = init->tz; }
    std::memset(out, 0, sizeof(*out));
    float initial_error = 0.f;
    for (int lvl = ap->first_level; lvl >= ap->last_level; lvl--) {                       // :1182
        // `cols`, `rows` = the reference's bookkeeping w_[lvl], h_[lvl] = size >> lvl (src/VISystem.cpp InitializePyramid): what the patch
        // builders bound the candidate points by; the Mats it indexes -- and tests the WARPED point against (:1299) -- have Camera::Update's
        // sizes (orc_half_pyramid_dims: up to one row / column more when a size does not halve exactly)
        const int cols = w >> lvl, rows = h >> lvl, N = n_cand[lvl];
        int32_t alw[5], alh[5]; orc_half_pyramid_dims(w, h, alw, alh);
        const int acols = alw[lvl], arows = alh[lvl];
        if (N < 0 || (N && (!gray1[lvl] || !gray2[lvl] || !gx1[lvl] || !gy1[lvl] || !cand1[lvl]))) return VIS_E_INVALID;
        const uint8_t* I1 = gray1[lvl]; const uint8_t* I2 = gray2[lvl];
        const float fx = K[lvl].fx, fy = K[lvl].fy, cx = K[lvl].cx, cy = K[lvl].cy, invfx = K[lvl].invfx, invfy = K[lvl].invfy;
        float error = 0.f, last_error = 50000.f;                                            // :1185-1186
        int k = 0, nres = 0;
        for (k = 0; k < ap->max_iterations; k++) {                                          // :1215
            float M[16];
            { vis_se3f p = {pose.q.x, pose.q.y, pose.q.z, pose.q.w, pose.t[0], pose.t[1], pose.t[2]}; orc_se3_matrix(&p, M); }
            std::vector<float> J, r;
            J.reserve((size_t)N * 6); r.reserve((size_t)N);
            double sumsq = 0;
            for (int i = 0; i < N; i++) {
                const float x1 = cand1[lvl][4 * i], y1 = cand1[lvl][4 * i + 1], z1 = cand1[lvl][4 * i + 2], w1 = cand1[lvl][4 * i + 3];
                // WarpFunctionSE3, :1495-1558: back-project, rigid * p (gemm: double products and sums), project
                const float X = ((x1 - cx) * invfx) * z1, Y = ((y1 - cy) * invfy) * z1;
                float P[4];
                for (int a = 0; a < 4; a++)
                    P[a] = (float)((((double)M[4 * a] * (double)X + (double)M[4 * a + 1] * (double)Y) + (double)M[4 * a + 2] * (double)z1) + (double)M[4 * a + 3] * (double)w1);
                float x2 = P[0] * fx; x2 = x2 / P[2]; x2 = x2 + cx;
                float y2 = P[1] * fy; y2 = y2 / P[2]; y2 = y2 + cy;
                x2 = x2 * P[3]; y2 = y2 * P[3];
                const float z2 = P[2];
                float inv_z2 = 1 / z2;
                if (!(y2 > 0 && y2 < arows && x2 > 0 && x2 < acols)) continue;               // :1299 `y2<image2.rows && x2<image2.cols`: the Mat's own size
                if (!(z2 != 0)) continue;                                                   // :1281
                if (inv_z2 < 0) inv_z2 = 0;                                                 // :1282-1283
                float Jw[2][6];
                Jw[0][0] = fx * inv_z2; Jw[0][1] = 0.f;
                Jw[0][2] = -(fx * x2 * inv_z2 * inv_z2) * ap->z_factor;
                Jw[0][3] = -(fx * x2 * y2 * inv_z2 * inv_z2);
                Jw[0][4] = (fx * (1 + x2 * x2 * inv_z2 * inv_z2));
                Jw[0][5] = -fx * y2 * inv_z2;
                Jw[1][0] = 0.f; Jw[1][1] = fy * inv_z2;
                Jw[1][2] = -(fy * y2 * inv_z2 * inv_z2) * ap->z_factor;
                Jw[1][3] = -(fy * (1 + y2 * y2 * inv_z2 * inv_z2));
                Jw[1][4] = fy * x2 * y2 * inv_z2 * inv_z2;
                Jw[1][5] = -fy * x2 * inv_z2;
                const int ix1 = (int)x1, iy1 = (int)y1;                                      // at<uchar>(y1, x1): float -> int
                if (ix1 < 0 || ix1 >= cols || iy1 < 0 || iy1 >= rows) continue;              // SPEC: the builders never emit such points
                int rx = (int)std::round(x2), ry = (int)std::round(y2);                      // :1305
                if (rx > acols - 1) rx = acols - 1;                                          // SPEC clamp (see header): to the Mat's own size
                if (ry > arows - 1) ry = arows - 1;
                const int intensity1 = I1[(size_t)iy1 * acols + ix1];
                const int intensity2 = I2[(size_t)ry * acols + rx];
                const float res = (float)(intensity2 - intensity1);
                const float jl0 = (float)gx1[lvl][(size_t)iy1 * acols + ix1], jl1 = (float)gy1[lvl][(size_t)iy1 * acols + ix1];
                for (int c = 0; c < 6; c++) J.push_back((float)((double)jl0 * (double)Jw[0][c] + (double)jl1 * (double)Jw[1][c]));   // Jl * Jw (gemm)
                r.push_back(res);
                sumsq += (double)res * (double)res;
            }
            nres = (int)r.size();
            if (nres == 0) break;                                                           // SPEC (see header)
            const float inv_num = (float)(1.0 / nres);                                      // :1331
            error = (float)((double)inv_num * sumsq);                                       // :1333-1334 (W = 1)
            if (k == 0) initial_error = error;                                              // lvl-local in effect: overwritten per level
            if (error >= last_error || k == ap->max_iterations - 1 || std::fabs(error - last_error) < ap->epsilon) break;   // :1341
            last_error = error;
            double S[27]; reduce_normal_equations(J, r, S);
            float A[36], b[6];
            { int s = 0; for (int a = 0; a < 6; a++) for (int c = a; c < 6; c++) { A[6 * a + c] = A[6 * c + a] = (float)S[s++]; } }
            for (int a = 0; a < 6; a++) b[a] = (float)(-S[21 + a]);                          // b = -J^T r (gemm alpha = -1)
            float Ainv[36];
            float Awork[36]; std::memcpy(Awork, A, sizeof(A));
            if (!lu_invert6(Awork, Ainv)) std::memset(Ainv, 0, sizeof(Ainv));               // cv::invert: singular -> zeros
            float delta[6];
            for (int a = 0; a < 6; a++) { double s = 0; for (int c = 0; c < 6; c++) s += (double)Ainv[6 * a + c] * (double)b[c]; delta[a] = (float)s; }
            pose = se3_mul(pose, se3_exp(delta));                                           // :1413
        }
        out->iterations[lvl] = k; out->error[lvl] = error; out->n_residuals[lvl] = nres;
    }
    out->initial_error = initial_error;
    out->pose.qx = pose.q.x; out->pose.qy = pose.q.y; out->pose.qz = pose.q.z; out->pose.qw = pose.q.w;
    out->pose.tx = pose.t[0]; out->pose.ty = pose.t[1]; out->pose.tz = pose.t[2];
    orc_se3_matrix(&out->pose, out->matrix);
    return VIS_OK;
}
