// align.hip -- VISystem::EstimatePoseFeatures on gfx950 (MI355X): batched Gauss-Newton photometric alignment.
//
// Replaces the pose step VISystemGPU::AddFrameGPU calls (/root/reference/src/VISystemGPU.cpp:167):
//   VISystem::EstimatePoseFeatures  /root/reference/src/VISystem.cpp:1113-1448
//   VISystem::WarpFunctionSE3       /root/reference/src/VISystem.cpp:1495-1558
//   VISystem::InitializePyramid     /root/reference/src/VISystem.cpp:1451-1493
// (SURVEY.md 8(f) N4).  It consumes what the N2 stage already produces on the device: the half pyramid, the Scharr
// gradients and the matched keypoints of the previous keyframe.
//
// Design: the algorithm is a strictly sequential chain of <= 4 levels x <= 10 iterations per frame pair, each iteration
// a reduction over a few thousand candidate pixels.  ONE persistent workgroup (4 waves) owns a pair for the whole
// chain: no host round trip and no kernel boundary between iterations, the pose lives in registers, the 27 sums of
// the normal equations are reduced with wave shuffles + 864 bytes of LDS, thread 0 solves the 6x6 system in LDS.
// Pairs are independent -> grid = pairs (1023 workgroups per 1024-frame batch).  In the batched path the candidate
// pixels of a level (the (2p+1)^2 patches around the matched keypoints, src/Camera.cpp:358-410) are generated once
// per level into LDS as packed (y << 16 | x) words instead of being streamed from a 16-byte-per-point list in HBM
// (95 KB per pair and level, re-read every iteration).  Image / gradient reads are scattered 1-5 byte gathers inside
// small patches: L2 hits after the first iteration.  Bound: FP64/FP32 VALU latency of one workgroup per pair.
//
// Arithmetic (operation order, float/double widths, summation tree) mirrors oracle/align.cpp exactly; the results are
// bit-identical to it (tests/test_align_gpu.py).
#include "vis_internal.h"
#include <cfloat>
#include <cstring>

#define DEV __device__ __forceinline__
#define HD __host__ __device__ inline              // the SE3 pieces also back the host-side vis_se3_* helpers of the adapters
#define AL_THREADS 256
#define AL_MAXKP 200                     // min(num_max_keypoints, 200), src/Camera.cpp:377

struct AlignLevel {
    const uint8_t* i1; const uint8_t* i2;          // level image of frame 0 of the batch (previous / current use frame offsets)
    const int16_t* gx; const int16_t* gy;
    const float* cand;                             // explicit list (x, y, z, w) rows, or nullptr (generated)
    size_t img_fstride, grad_fstride;              // elements between consecutive frames
    int rowstride, cols, rows, ncand;              // cols, rows: the reference's bookkeeping w_[lvl], h_[lvl] = size >> lvl (bounds of the coordinates)
    int acols, arows;                              // the level's own size (Camera::Update: cvRound(size * 0.5)): stride of the dense gradients, clamp of the rounded index
    float fx, fy, cx, cy, invfx, invfy;
};
struct AlignArgs {
    AlignLevel lv[5];
    int first, last, max_it; float eps, zf;
    const float* pts; const int32_t* npts; int max_pts;   // generated mode: matched keypoints of the previous frame per pair
    const vis_se3f* init; vis_align_result* out;
    int f1_off, f2_off, out_off;                          // frame index of image 1 / image 2 / result slot = pair + offset
};

// ---- Sophus::SE3f pieces (see oracle/align.cpp for the citations) -------------------------------------------
struct Quat { float w, x, y, z; };
struct Se3 { Quat q; float t[3]; };

// deterministic double sin/cos, identical to detect.hip's / the oracle's sincos_det
HD void al_sincos(double x, double* s, double* c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;
    const double PIO2_LO = 6.07710050650619224932e-11;
    const double kd = rint(x * TWO_OVER_PI);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    const double sr = r + (r * z) * ps;
    const double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    const double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    switch (k & 3) {
        case 0: *s = sr;  *c = cr;  break;
        case 1: *s = cr;  *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}
HD float sin_det(float a) { double s, c; al_sincos((double)a, &s, &c); return (float)s; }
HD float cos_det(float a) { double s, c; al_sincos((double)a, &s, &c); return (float)c; }

HD Quat qmul(const Quat& a, const Quat& b) {
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
HD void qrot(const Quat& q, const float (&v)[3], float (&out)[3]) {
    float uv0 = q.y * v[2] - q.z * v[1], uv1 = q.z * v[0] - q.x * v[2], uv2 = q.x * v[1] - q.y * v[0];
    uv0 += uv0; uv1 += uv1; uv2 += uv2;
    const float c0 = q.y * uv2 - q.z * uv1, c1 = q.z * uv0 - q.x * uv2, c2 = q.x * uv1 - q.y * uv0;
    out[0] = v[0] + q.w * uv0 + c0;
    out[1] = v[1] + q.w * uv1 + c1;
    out[2] = v[2] + q.w * uv2 + c2;
}
HD void qmat(const Quat& q, float (&R)[9]) {
    const float tx = 2.f * q.x, ty = 2.f * q.y, tz = 2.f * q.z;
    const float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}
HD Se3 se3_exp(const float (&a)[6]) {
    const float o0 = a[3], o1 = a[4], o2 = a[5];
    const float theta_sq = o0 * o0 + o1 * o1 + o2 * o2;
    const float theta = sqrtf(theta_sq);
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
    r.q.w = real; r.q.x = imag * o0; r.q.y = imag * o1; r.q.z = imag * o2;
    const float O[9] = {0.f, -o2, o1, o2, 0.f, -o0, -o1, o0, 0.f};
    float O2[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    float V[9];
    if (theta < eps) qmat(r.q, V);
    else {
        const float ca = (1.f - cos_det(theta)) / theta_sq;
        const float cb = (theta - sin_det(theta)) / (theta_sq * theta);
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + ca * O[i] + cb * O2[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) r.t[i] = V[3 * i] * a[0] + V[3 * i + 1] * a[1] + V[3 * i + 2] * a[2];
    return r;
}
HD Se3 se3_mul(const Se3& a, const Se3& b) {
    Se3 r = a;
    float rt[3];
    qrot(a.q, b.t, rt);
    r.t[0] = a.t[0] + rt[0]; r.t[1] = a.t[1] + rt[1]; r.t[2] = a.t[2] + rt[2];
    r.q = qmul(a.q, b.q);
    const float sn = r.q.w * r.q.w + r.q.x * r.q.x + r.q.y * r.q.y + r.q.z * r.q.z;
    if (sn != 1.f) { const float s = 2.f / (1.f + sn); r.q.w *= s; r.q.x *= s; r.q.y *= s; r.q.z *= s; }
    return r;
}

// hal::LU32f as cv::invert(DECOMP_LU) uses it: A (6x6, row-major, LDS) -> B = inverse; false = singular.  The matrices are
// held in registers for the factorisation (fully unrolled; a row exchange with the runtime pivot row is a chain of selects):
// the same operations in the same order as the in-place version, without ~800 dependent LDS round trips of one thread.
__device__ bool lu_invert6(const float* Ain, float* Bout) {
    constexpr int n = 6;
    float A[n][n], B[n][n];
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < n; j++) { A[i][j] = Ain[i * n + j]; B[i][j] = i == j ? 1.f : 0.f; }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < n; i++) {
        int k = i; float best = fabsf(A[i][i]);
#pragma unroll
        for (int j = i + 1; j < n; j++) { const float v = fabsf(A[j][i]); if (v > best) { best = v; k = j; } }
        if (best < FLT_EPSILON * 10) ok = false;
#pragma unroll
        for (int j = i + 1; j < n; j++)
            if (k == j) {
#pragma unroll
                for (int c = i; c < n; c++) { const float t = A[i][c]; A[i][c] = A[j][c]; A[j][c] = t; }
#pragma unroll
                for (int c = 0; c < n; c++) { const float t = B[i][c]; B[i][c] = B[j][c]; B[j][c] = t; }
            }
        const float d = -1.f / A[i][i];
#pragma unroll
        for (int j = i + 1; j < n; j++) {
            const float alpha = A[j][i] * d;
#pragma unroll
            for (int c = i + 1; c < n; c++) A[j][c] += alpha * A[i][c];
#pragma unroll
            for (int c = 0; c < n; c++) B[j][c] += alpha * B[i][c];
        }
    }
#pragma unroll
    for (int i = n - 1; i >= 0; i--)
#pragma unroll
        for (int j = 0; j < n; j++) {
            float s_ = B[i][j];
#pragma unroll
            for (int k = i + 1; k < n; k++) s_ -= A[i][k] * B[k][j];
            B[i][j] = s_ / A[i][i];
        }
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < n; j++) Bout[i * n + j] = B[i][j];
    return ok;
}

// sum over the 64 lanes of a wave in the oracle's tree order (stride 32, 16, ... 1): lane 0 holds the result
DEV double wave_tree_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
DEV unsigned long long wave_tree_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// One workgroup per frame pair; GEN = candidate pixels generated from the matched keypoints (batched path),
// otherwise read from an explicit (x, y, z, w) list (single-pair entry point: Frame::candidatePoints as the caller holds it).
template <bool GEN>
__global__ __launch_bounds__(AL_THREADS) void k_align(AlignArgs G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* plist = reinterpret_cast<uint32_t*>(smem);                 // GEN: packed candidate pixels of the level
    __shared__ double s_red[4][28];
    __shared__ unsigned long long s_sq[4];
    __shared__ int s_cnt[4];
    __shared__ float s_A[36], s_Ainv[36], s_delta[6];
    __shared__ int s_pre[AL_THREADS + 1];
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pair = blockIdx.x;
    const int f1 = pair + G.f1_off, f2 = pair + G.f2_off;
    Se3 pose = {{1.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (G.init) {
        const vis_se3f in = G.init[pair + G.out_off];
        pose.q.w = in.qw; pose.q.x = in.qx; pose.q.y = in.qy; pose.q.z = in.qz; pose.t[0] = in.tx; pose.t[1] = in.ty; pose.t[2] = in.tz;
    }
    vis_align_result res;
    if (tid == 0) {
        for (int l = 0; l < 5; l++) { res.error[l] = 0.f; res.iterations[l] = 0; res.n_residuals[l] = 0; }
        res.initial_error = 0.f;
    }
    float initial_error = 0.f;
    const int m = GEN ? min(min(G.npts[pair + G.out_off], G.max_pts), AL_MAXKP) : 0;
    const float* kp = GEN ? G.pts + (size_t)(pair + G.out_off) * G.max_pts * 2 : nullptr;
#pragma unroll 1
    for (int lvl = G.first; lvl >= G.last; lvl--) {
        const AlignLevel V = G.lv[lvl];
        const int cols = V.cols, rows = V.rows;
        int N = V.ncand;
        if (GEN) {
            // Camera::ObtainPatchesPointsPreviousFrame (src/Camera.cpp:358-410) for this level, into LDS
            const int patch_size = lvl == 1 ? 3 : (lvl == 2 ? 2 : 5);
            const int sp = patch_size - 1 / 2;                                      // :376, integer 1/2 == 0
            const float factor_lvl = (float)(1.0 / (double)(1 << lvl));
            float x = 0, y = 0; int cnt = 0;
            if (tid < m) {
                x = (float)(((double)kp[2 * tid] + 0.5) * (double)factor_lvl - 0.5);
                y = (float)(((double)kp[2 * tid + 1] + 0.5) * (double)factor_lvl - 0.5);
                for (int i = (int)(x - (float)sp); (float)i <= x + (float)sp; i++)
                    for (int j = (int)(y - (float)sp); (float)j <= y + (float)sp; j++)
                        if (i > 0 && i < cols && j > 0 && j < rows) cnt++;
            }
            __syncthreads();                                                         // previous level done with plist / s_pre
            // exclusive prefix of the per-keypoint counts: wave scans by shuffles + the three wave totals (one thread walking
            // the 256 entries was 256 dependent LDS round trips per level)
            {
                int v = cnt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
                if (lane == 63) s_cnt[wv] = v;
                __syncthreads();
                int base = 0;
                for (int w = 0; w < wv; w++) base += s_cnt[w];
                s_pre[tid + 1] = base + v;
                if (tid == 0) s_pre[0] = 0;
                __syncthreads();
            }
            if (tid < m) {
                int o = s_pre[tid];
                for (int i = (int)(x - (float)sp); (float)i <= x + (float)sp; i++)
                    for (int j = (int)(y - (float)sp); (float)j <= y + (float)sp; j++)
                        if (i > 0 && i < cols && j > 0 && j < rows) plist[o++] = ((uint32_t)j << 16) | (uint32_t)i;
            }
            N = s_pre[AL_THREADS];
            __syncthreads();
        }
        const uint8_t* I1 = V.i1 + (size_t)f1 * V.img_fstride;
        const uint8_t* I2 = V.i2 + (size_t)f2 * V.img_fstride;
        const int16_t* GX = V.gx + (size_t)f1 * V.grad_fstride;
        const int16_t* GY = V.gy + (size_t)f1 * V.grad_fstride;
        const float fx = V.fx, fy = V.fy, cx = V.cx, cy = V.cy, invfx = V.invfx, invfy = V.invfy;
        float error = 0.f, last_error = 50000.f;
        int k = 0, nres = 0;
#pragma unroll 1
        for (k = 0; k < G.max_it; k++) {
            float R[9]; qmat(pose.q, R);
            // rows 0..2 of pose.matrix(); row 3 = (0 0 0 1)
            const double m00 = R[0], m01 = R[1], m02 = R[2], m03 = pose.t[0];
            const double m10 = R[3], m11 = R[4], m12 = R[5], m13 = pose.t[1];
            const double m20 = R[6], m21 = R[7], m22 = R[8], m23 = pose.t[2];
            double S[27];
#pragma unroll
            for (int s = 0; s < 27; s++) S[s] = 0.0;
            unsigned long long sumsq = 0; int cnt = 0;
            // Four candidates per thread per round: first the warp of all four and their image / gradient loads (clamped
            // addresses for the ones that will be skipped), then the accumulation in the original order i, i + T, i + 2T,
            // i + 3T -- the per-thread sums see the same operations in the same order, but a round costs one memory round
            // trip instead of four.
#pragma unroll 1
            for (int i0 = tid; i0 < N; i0 += 4 * AL_THREADS) {
                float x2a[4], y2a[4], iza[4]; bool ok[4]; int i1v[4], i2v[4]; float gxv[4], gyv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = i0 + u * AL_THREADS;
                    ok[u] = false; x2a[u] = y2a[u] = iza[u] = 0.f;
                    size_t o1 = 0, o2 = 0, go = 0;
                    if (i < N) {
                        float x1, y1, z1, w1;
                        if (GEN) { const uint32_t pw = plist[i]; x1 = (float)(pw & 0xFFFFu); y1 = (float)(pw >> 16); z1 = 1.f; w1 = 1.f; }
                        else { const float4 c = reinterpret_cast<const float4*>(V.cand)[i]; x1 = c.x; y1 = c.y; z1 = c.z; w1 = c.w; }
                        const float X = ((x1 - cx) * invfx) * z1, Y = ((y1 - cy) * invfy) * z1;
                        const float P0 = (float)(((m00 * (double)X + m01 * (double)Y) + m02 * (double)z1) + m03 * (double)w1);
                        const float P1 = (float)(((m10 * (double)X + m11 * (double)Y) + m12 * (double)z1) + m13 * (double)w1);
                        const float P2 = (float)(((m20 * (double)X + m21 * (double)Y) + m22 * (double)z1) + m23 * (double)w1);
                        const float P3 = (float)(((0.0 * (double)X + 0.0 * (double)Y) + 0.0 * (double)z1) + 1.0 * (double)w1);
                        float x2 = P0 * fx; x2 = x2 / P2; x2 = x2 + cx;
                        float y2 = P1 * fy; y2 = y2 / P2; y2 = y2 + cy;
                        x2 = x2 * P3; y2 = y2 * P3;
                        const float z2 = P2;
                        float inv_z2 = 1 / z2;
                        const int ix1 = (int)x1, iy1 = (int)y1;
                        // the warped point is tested against the size of the Mat it indexes (src/VISystem.cpp:1299: image2.rows / image2.cols
                        // = Camera::Update's level size, up to one row / column MORE than the bookkeeping `size >> lvl`)
                        bool v = (y2 > 0 && y2 < V.arows && x2 > 0 && x2 < V.acols) && (z2 != 0);
                        if (inv_z2 < 0) inv_z2 = 0;
                        v = v && !(ix1 < 0 || ix1 >= cols || iy1 < 0 || iy1 >= rows);
                        if (v) {
                            int rx = (int)roundf(x2), ry = (int)roundf(y2);
                            if (rx > V.acols - 1) rx = V.acols - 1;
                            if (ry > V.arows - 1) ry = V.arows - 1;
                            o1 = (size_t)iy1 * V.rowstride + ix1; o2 = (size_t)ry * V.rowstride + rx; go = (size_t)iy1 * V.acols + ix1;   // gradients are dense
                        }
                        ok[u] = v; x2a[u] = x2; y2a[u] = y2; iza[u] = inv_z2;
                    }
                    i1v[u] = I1[o1]; i2v[u] = I2[o2]; gxv[u] = (float)GX[go]; gyv[u] = (float)GY[go];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!ok[u]) continue;
                    const float x2 = x2a[u], y2 = y2a[u], inv_z2 = iza[u];
                    float Jw0[6], Jw1[6];
                    Jw0[0] = fx * inv_z2; Jw0[1] = 0.f;
                    Jw0[2] = -(fx * x2 * inv_z2 * inv_z2) * G.zf;
                    Jw0[3] = -(fx * x2 * y2 * inv_z2 * inv_z2);
                    Jw0[4] = (fx * (1 + x2 * x2 * inv_z2 * inv_z2));
                    Jw0[5] = -fx * y2 * inv_z2;
                    Jw1[0] = 0.f; Jw1[1] = fy * inv_z2;
                    Jw1[2] = -(fy * y2 * inv_z2 * inv_z2) * G.zf;
                    Jw1[3] = -(fy * (1 + y2 * y2 * inv_z2 * inv_z2));
                    Jw1[4] = fy * x2 * y2 * inv_z2 * inv_z2;
                    Jw1[5] = -fy * x2 * inv_z2;
                    const int ri = i2v[u] - i1v[u];
                    const float resf = (float)ri;
                    const float jl0 = gxv[u], jl1 = gyv[u];
                    double J[6];
#pragma unroll
                    for (int c = 0; c < 6; c++) J[c] = (double)(float)((double)jl0 * (double)Jw0[c] + (double)jl1 * (double)Jw1[c]);
                    int s = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int b = a; b < 6; b++) S[s++] += J[a] * J[b];
#pragma unroll
                    for (int a = 0; a < 6; a++) S[21 + a] += J[a] * (double)resf;
                    sumsq += (unsigned long long)(ri * ri);
                    cnt++;
                }
            }
            // reduction in the oracle's order: inside a wave strides 32..1, then (W0 + W1) + (W2 + W3)
#pragma unroll
            for (int s = 0; s < 27; s++) { const double v = wave_tree_sum(S[s]); if (lane == 0) s_red[wv][s] = v; }
            { const unsigned long long q = wave_tree_sum_u64(sumsq); if (lane == 0) s_sq[wv] = q; }
            { int c = cnt; for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64); if (lane == 0) s_cnt[wv] = c; }
            __syncthreads();
            nres = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
            const unsigned long long tsq = (s_sq[0] + s_sq[1]) + (s_sq[2] + s_sq[3]);
            if (nres == 0) { __syncthreads(); break; }
            const float inv_num = (float)(1.0 / nres);
            error = (float)((double)inv_num * (double)tsq);
            if (k == 0) initial_error = error;
            if (error >= last_error || k == G.max_it - 1 || fabsf(error - last_error) < G.eps) { __syncthreads(); break; }
            last_error = error;
            if (tid == 0) {
                float b[6];
                int s = 0;
                for (int a = 0; a < 6; a++)
                    for (int c = a; c < 6; c++) { const float v = (float)((s_red[0][s] + s_red[1][s]) + (s_red[2][s] + s_red[3][s])); s_A[6 * a + c] = v; s_A[6 * c + a] = v; s++; }
                for (int a = 0; a < 6; a++) b[a] = (float)(-((s_red[0][21 + a] + s_red[1][21 + a]) + (s_red[2][21 + a] + s_red[3][21 + a])));
                if (!lu_invert6(s_A, s_Ainv)) for (int i = 0; i < 36; i++) s_Ainv[i] = 0.f;
                for (int a = 0; a < 6; a++) { double acc = 0; for (int c = 0; c < 6; c++) acc += (double)s_Ainv[6 * a + c] * (double)b[c]; s_delta[a] = (float)acc; }
            }
            __syncthreads();
            float delta[6];
#pragma unroll
            for (int a = 0; a < 6; a++) delta[a] = s_delta[a];
            pose = se3_mul(pose, se3_exp(delta));
            __syncthreads();                                                        // s_red / s_delta are rewritten next iteration
        }
        if (tid == 0) { res.iterations[lvl] = k; res.error[lvl] = error; res.n_residuals[lvl] = nres; }
    }
    if (tid == 0) {
        res.initial_error = initial_error;
        res.pose.qx = pose.q.x; res.pose.qy = pose.q.y; res.pose.qz = pose.q.z; res.pose.qw = pose.q.w;
        res.pose.tx = pose.t[0]; res.pose.ty = pose.t[1]; res.pose.tz = pose.t[2];
        float R[9]; qmat(pose.q, R);
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) res.matrix[4 * i + j] = R[3 * i + j]; res.matrix[4 * i + 3] = pose.t[i]; }
        res.matrix[12] = res.matrix[13] = res.matrix[14] = 0.f; res.matrix[15] = 1.f;
        G.out[pair + G.out_off] = res;
    }
    (void)s_ok;
}

// ------------------------------------------------------------------------------------------------ host side
// Sophus::SE3f value operations for the host adapters (VISystem::Track composes poses, src/VISystem.cpp:1567-1635): the same
// functions the kernel runs, evaluated on the host.
static Se3 to_se3(const vis_se3f& a) { Se3 r; r.q.w = a.qw; r.q.x = a.qx; r.q.y = a.qy; r.q.z = a.qz; r.t[0] = a.tx; r.t[1] = a.ty; r.t[2] = a.tz; return r; }
static void from_se3(const Se3& e, vis_se3f* o) { o->qx = e.q.x; o->qy = e.q.y; o->qz = e.q.z; o->qw = e.q.w; o->tx = e.t[0]; o->ty = e.t[1]; o->tz = e.t[2]; }
extern "C" void vis_se3_exp(const float a[6], vis_se3f* out) { const float v[6] = {a[0], a[1], a[2], a[3], a[4], a[5]}; from_se3(se3_exp(v), out); }
extern "C" void vis_se3_mul(const vis_se3f* a, const vis_se3f* b, vis_se3f* out) { from_se3(se3_mul(to_se3(*a), to_se3(*b)), out); }
extern "C" void vis_se3_matrix(const vis_se3f* a, float M[16]) {
    float R[9]; qmat(to_se3(*a).q, R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[4 * i + j] = R[3 * i + j];
    M[3] = a->tx; M[7] = a->ty; M[11] = a->tz; M[12] = M[13] = M[14] = 0.f; M[15] = 1.f;
}
// SE3(Matrix3 R, Point t): Eigen's rotation-matrix -> quaternion conversion
extern "C" void vis_se3_from_rt(const float R[9], const float t[3], vis_se3f* out) {
    float qw, v[3];
    float tr = R[0] + R[4] + R[8];
    if (tr > 0.f) {
        tr = sqrtf(tr + 1.f);
        qw = 0.5f * tr;
        tr = 0.5f / tr;
        v[0] = (R[7] - R[5]) * tr; v[1] = (R[2] - R[6]) * tr; v[2] = (R[3] - R[1]) * tr;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        tr = sqrtf(R[4 * i] - R[4 * j] - R[4 * k] + 1.f);
        v[i] = 0.5f * tr;
        tr = 0.5f / tr;
        qw = (R[3 * k + j] - R[3 * j + k]) * tr;
        v[j] = (R[3 * j + i] + R[3 * i + j]) * tr;
        v[k] = (R[3 * k + i] + R[3 * i + k]) * tr;
    }
    out->qx = v[0]; out->qy = v[1]; out->qz = v[2]; out->qw = qw; out->tx = t[0]; out->ty = t[1]; out->tz = t[2];
}

extern "C" void vis_default_align_params(vis_align_params* ap) {
    if (!ap) return;
    std::memset(ap, 0, sizeof(*ap));
    ap->fx = 458.654f; ap->fy = 457.296f; ap->cx = 367.215f; ap->cy = 248.375f;     // calibrationEUROC.xml:20
    ap->first_level = 3; ap->last_level = 0; ap->max_iterations = 10;                // src/VISystem.cpp:1117,1119,1120
    ap->epsilon = 0.001f; ap->z_factor = 0.002f;                                      // :1115,1121
}

static int check_align_params(const vis_align_params* ap, int w, int h) {
    if (!ap || ap->first_level < ap->last_level || ap->first_level > 4 || ap->last_level < 0 || ap->max_iterations < 1) return VIS_E_INVALID;
    if (w < 16 || h < 16 || w > 4095 || h > 4095 || !(ap->fx > 0) || !(ap->fy > 0)) return VIS_E_INVALID;
    return VIS_OK;
}

// VISystem::InitializePyramid, src/VISystem.cpp:1451-1493 (same IEEE operations as the oracle)
static void fill_level_intrinsics(const vis_align_params& ap, AlignArgs& G) {
    float fx[5], fy[5], cx[5], cy[5];
    fx[0] = ap.fx; fy[0] = ap.fy; cx[0] = ap.cx; cy[0] = ap.cy;
    for (int l = 1; l < 5; l++) {
        fx[l] = (float)((double)fx[l - 1] * 0.5);
        fy[l] = (float)((double)fy[l - 1] * 0.5);
        cx[l] = (float)(((double)cx[0] + 0.5) / (double)(1 << l) - 0.5);
        cy[l] = (float)(((double)cy[0] + 0.5) / (double)(1 << l) - 0.5);
    }
    for (int l = 0; l < 5; l++) {
        G.lv[l].fx = fx[l]; G.lv[l].fy = fy[l]; G.lv[l].cx = cx[l]; G.lv[l].cy = cy[l];
        G.lv[l].invfx = 1.f / fx[l]; G.lv[l].invfy = 1.f / fy[l];
    }
    G.first = ap.first_level; G.last = ap.last_level; G.max_it = ap.max_iterations; G.eps = ap.epsilon; G.zf = ap.z_factor;
}

extern "C" int vis_estimate_pose_features(vis_ctx* ctx, const vis_align_params* ap, int w, int h,
                                          const uint8_t* const gray1[5], const uint8_t* const gray2[5],
                                          const int16_t* const gx1[5], const int16_t* const gy1[5],
                                          const float* const cand1[5], const int32_t n_cand[5],
                                          const vis_se3f* init, vis_align_result* out) {
    if (!ctx || !out || !gray1 || !gray2 || !gx1 || !gy1 || !cand1 || !n_cand) return VIS_E_INVALID;
    int rc = check_align_params(ap, w, h);
    if (rc) return rc;
    (void)hipSetDevice(ctx->device);
    size_t need = 4096;
    int alw[5], alh[5]; vis_half_dims(w, h, alw, alh);
    for (int l = ap->last_level; l <= ap->first_level; l++) {
        const int N = n_cand[l];
        if (N < 0 || N > (1 << 24)) return VIS_E_INVALID;
        if (N && (!gray1[l] || !gray2[l] || !gx1[l] || !gy1[l] || !cand1[l])) return VIS_E_INVALID;
        const size_t px = (size_t)alw[l] * alh[l];
        need += 6 * px + (size_t)N * 16 + 6 * 256;
    }
    rc = vis_ensure_scratch(ctx, need + sizeof(vis_se3f) + sizeof(vis_align_result) + 1024);
    if (!rc) rc = vis_ensure_pin(ctx, need + sizeof(vis_se3f) + sizeof(vis_align_result) + 4096);     // (the caller's arrays are pageable: through the pinned block, one wait per call)
    if (rc) return rc;
    HostStage hs(ctx);
    Carver cv{(char*)ctx->d_scratch, 0};
    AlignArgs G; std::memset(&G, 0, sizeof(G));
    fill_level_intrinsics(*ap, G);
    hipStream_t st = ctx->stream;
    for (int l = ap->last_level; l <= ap->first_level; l++) {
        const int N = n_cand[l];
        const int cols = w >> l, rows = h >> l;
        const size_t px = (size_t)alw[l] * alh[l];
        AlignLevel& V = G.lv[l];
        V.cols = cols; V.rows = rows; V.acols = alw[l]; V.arows = alh[l]; V.rowstride = alw[l]; V.ncand = N; V.img_fstride = 0; V.grad_fstride = 0;
        if (!N) continue;
        uint8_t* d1 = cv.take<uint8_t>(px); uint8_t* d2 = cv.take<uint8_t>(px);
        int16_t* dgx = cv.take<int16_t>(px); int16_t* dgy = cv.take<int16_t>(px);
        float* dc = cv.take<float>((size_t)N * 4);
        hs.up(d1, gray1[l], px); hs.up(d2, gray2[l], px);
        hs.up(dgx, gx1[l], px * 2); hs.up(dgy, gy1[l], px * 2);
        hs.up(dc, cand1[l], (size_t)N * 16);
        V.i1 = d1; V.i2 = d2; V.gx = dgx; V.gy = dgy; V.cand = dc;
    }
    vis_se3f* d_init = nullptr;
    if (init) { d_init = cv.take<vis_se3f>(1); hs.up(d_init, init, sizeof(vis_se3f)); }
    hs.flush_ups();
    vis_align_result* d_out = cv.take<vis_align_result>(1);
    G.init = d_init; G.out = d_out; G.f1_off = 0; G.f2_off = 0; G.out_off = 0;
    hipLaunchKernelGGL(k_align<false>, dim3(1), dim3(AL_THREADS), 0, st, G);
    HIPCHK(ctx, hipGetLastError());
    const void* h_out = hs.down(d_out, sizeof(vis_align_result));
    rc = hs.wait();
    if (rc) return rc;
    std::memcpy(out, h_out, sizeof(vis_align_result));
    return VIS_OK;
}

extern "C" int vis_align_batch(vis_ctx* ctx, const vis_align_params* ap, const uint8_t* d_frames, int w, int h, int stride, int n,
                               const uint8_t* d_gray, const int16_t* d_gx, const int16_t* d_gy,
                               const float* d_pts, const int32_t* d_npts, int max_pts,
                               const vis_se3f* d_init, vis_align_result* d_out) {
    if (!ctx || !d_frames || !d_gray || !d_gx || !d_gy || !d_pts || !d_npts || !d_out) return VIS_E_INVALID;
    int rc = check_align_params(ap, w, h);
    if (rc) return rc;
    if (stride < w || n < 1 || max_pts < 1) { ctx->err = "vis_align_batch: stride >= w, n >= 1, max_pts >= 1"; return VIS_E_INVALID; }
    (void)hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    {   // frame 0 has no predecessor in this batch: its record is cleared (by a kernel of the library, not by the runtime's fill)
        static_assert(sizeof(vis_align_result) % 4 == 0, "cleared in dwords");
        void* dsts[1] = {d_out}; const void* srcs[1] = {nullptr}; const size_t bytes[1] = {sizeof(vis_align_result)};
        const int rc0 = launch_copy_jobs(ctx, st, 1, dsts, srcs, bytes);
        if (rc0) return rc0;
    }
    if (n < 2) return VIS_OK;
    AlignArgs G; std::memset(&G, 0, sizeof(G));
    fill_level_intrinsics(*ap, G);
    const size_t fe = vis_grad_frame_elems(w, h);
    int alw[5], alh[5]; vis_half_dims(w, h, alw, alh);
    size_t off = 0;
    for (int l = 0; l < 5; l++) {
        AlignLevel& V = G.lv[l];
        V.cols = w >> l; V.rows = h >> l; V.acols = alw[l]; V.arows = alh[l]; V.ncand = 0; V.cand = nullptr;
        if (l == 0) { V.i1 = d_frames; V.i2 = d_frames; V.rowstride = stride; V.img_fstride = (size_t)stride * h; }
        else { V.i1 = d_gray + off; V.i2 = d_gray + off; V.rowstride = V.acols; V.img_fstride = fe; }
        V.gx = d_gx + off; V.gy = d_gy + off; V.grad_fstride = fe;
        off += (size_t)V.acols * V.arows;
    }
    G.pts = d_pts; G.npts = d_npts; G.max_pts = max_pts; G.init = d_init; G.out = d_out;
    G.f1_off = 0; G.f2_off = 1; G.out_off = 1;                                       // workgroup q = pair (frame q -> frame q+1), result slot q+1
    const int used = std::min(max_pts, AL_MAXKP);
    const size_t lds = (size_t)used * 121 * 4;                                       // largest patch: (2*5+1)^2 pixels per keypoint
    if (lds > 65536 - 4096) HIPCHK(ctx, hipFuncSetAttribute((const void*)k_align<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_align<true>, dim3(n - 1), dim3(AL_THREADS), lds, st, G);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}

extern "C" int vis_batch_align(vis_ctx* ctx, const vis_align_params* ap, const uint8_t* d_frames, int n,
                               const uint8_t* d_gray, const int16_t* d_gx, const int16_t* d_gy,
                               const vis_se3f* d_init, vis_align_result* d_out) {
    if (!ctx) return VIS_E_INVALID;
    Plan* pl = ctx->batch;
    if (!pl || pl->last_n < 1 || n != pl->last_n) return VIS_E_STATE;
    if (ctx->p.pose_input != VIS_POSE_GOOD) return VIS_E_STATE;                       // needs the grid-filtered matches in d_p1
    bool plan_set = false;
    if (!d_gray && !d_gx && !d_gy) {                                                   // the plan's own gradients (VIS_STAGE_GRADIENT of the last vis_batch_run)
        if (!pl->grad_valid) { ctx->err = "vis_batch_align: no gradient buffers given and the last vis_batch_run had no VIS_STAGE_GRADIENT"; return VIS_E_STATE; }
        d_gray = pl->d_half; d_gx = pl->d_gx; d_gy = pl->d_gy; plan_set = true;
    }
    // One persistent workgroup per pair, a chain of dependent iterations: latency-bound work that fits beside the next batch's
    // detect chain.  It runs on the pose stream, behind (a) everything queued on the context's stream so far -- the gradients of
    // these frames -- and (b) the matcher of the last batch; what may not overtake it (the next filter rewriting the matched
    // points, the next vis_gradient_batch, the feeder's next copy into these frames) waits for ev_align_done.
    (void)hipSetDevice(ctx->device);
    hipStream_t sA = ctx->stream, sP = ctx->pose_stream;
    HIPCHK(ctx, hipEventRecord(ctx->ev_align_fork, sA));
    HIPCHK(ctx, hipStreamWaitEvent(sP, ctx->ev_align_fork, 0));
    if (pl->match_pending[pl->last_base / pl->rec_per_set]) HIPCHK(ctx, hipStreamWaitEvent(sP, ctx->ev_match_done[pl->last_base / pl->rec_per_set], 0));
    // d_p1 = the matched keypoints of the query frame of every pair (getGoodMatches, src/Matcher.cpp:295-303): pair i = (frame i-1, frame i)
    ctx->stream = sP;
    const int rc = vis_align_batch(ctx, ap, d_frames, pl->w, pl->h, pl->stride, n, d_gray, d_gx, d_gy, pl->d_p1, pl->d_ngood, pl->root * pl->root, d_init, d_out);
    ctx->stream = sA;
    if (rc) return rc;
    ctx->align_k ^= 1;
    ctx->ev_align_done = ctx->ev_align_done2[ctx->align_k];                           // (two events in turn: the one before stays valid for the set it guards)
    HIPCHK(ctx, hipEventRecord(ctx->ev_align_done, sP));
    ctx->align_pending = true;
    // the side stream refills a gradient set two steps on: it waits for the last alignment that read THAT set.  Decided by pointer
    // identity, not by how the caller got the pointers: the ones vis_batch_gradients() / vis_batch_half_pyramid() hand out are the
    // plan's sets too (valid until the next vis_batch_run), and a caller passing them explicitly needs the same ordering
    (void)plan_set;
    for (int s_ = 0; s_ < 2; s_++)
        if ((pl->d_half_set[s_] && d_gray == pl->d_half_set[s_]) || (pl->d_gx_set[s_] && d_gx == pl->d_gx_set[s_]) || (pl->d_gy_set[s_] && d_gy == pl->d_gy_set[s_]))
            pl->grad_reader[s_] = ctx->ev_align_done;
    pl->mo_align[pl->last_cur] = ctx->ev_align_done;                                 // it read the matched points of the last step's matcher-output set
    return VIS_OK;
}
