// oracle/orb.cpp -- CPU ORACLE (test infrastructure only; see vis_oracle.h header).
//
// Restates what cv::ORB::detectAndCompute computes for the reference call
//   detector->detectAndCompute(currentFrame->grayImage[0], Mat(), keypoints, descriptors)
// (/root/reference/src/Camera.cpp:84-92, ORB created at src/Camera.cpp:125-130 with OpenCV
// defaults; GPU twin src/CameraGPU.cpp:99-103) and Camera::Update's half pyramid
// (src/Camera.cpp:63-72).  OpenCV 3.2 itself is absent: PARITY UNPINNED vs real OpenCV.
// Section numbers "A.1 item k" refer to SURVEY.md Appendix A.
#include "vis_oracle.h"
#include "oracle_internal.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace orc {

// ---- deterministic double-precision sin/cos (spec shared with the HIP kernels) -----
// Cody-Waite reduction by pi/2 + fdlibm kernel polynomials, only IEEE + - * (no FMA):
// the same operation sequence on CPU and GPU gives the same bits.  Stands in for
// (float)cos(angle), (float)sin(angle) in computeOrbDescriptors (A.1 item 8).
void sincos_det(double x, double* s, double* c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;  // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;  // pi/2 - PIO2_HI
    double kd = std::nearbyint(x * TWO_OVER_PI);
    int k = (int)kd;
    double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))));
    double sr = r + (r * z) * ps;
    double pc = C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))));
    double cr = (1.0 - 0.5 * z) + (z * z) * pc;
    switch (k & 3) {
        case 0: *s = sr;  *c = cr;  break;
        case 1: *s = cr;  *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

static inline int cv_round(double v) { return (int)std::lrint(v); }          // cvRound
static inline int cv_floor_f(float v) { return (int)std::floor(v); }          // cvFloor

// ---- A.1 items 2,3: level geometry ----------------------------------------------------
void level_geometry(const vis_params& p, int w, int h, std::vector<LevelGeom>& g) {
    g.resize(p.nlevels);
    const double scaleFactor = (double)p.scale_factor;   // ORB_Impl stores the float arg in a double
    for (int l = 0; l < p.nlevels; l++) {
        float s = (float)std::pow(scaleFactor, (double)l);          // getScale()
        g[l].scale = s;
        g[l].w = cv_round((double)((float)w / s));                  // Size(cvRound(cols/scale), ..)
        g[l].h = cv_round((double)((float)h / s));
    }
    float factor = (float)(1.0 / scaleFactor);
    float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; l++) {
        g[l].quota = cv_round((double)nd);
        sum += g[l].quota;
        nd *= factor;
    }
    g[p.nlevels - 1].quota = std::max(p.nfeatures - sum, 0);
}

// ---- A.1 item 2: cv::resize INTER_LINEAR, 8-bit fixed point ------------------------------
void resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                   uint8_t* dst, int dw, int dh, int dstride) {
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        ialpha[2 * dx]     = (short)cv_round((double)(c0 * 2048));   // saturate_cast<short>
        ialpha[2 * dx + 1] = (short)cv_round((double)(c1 * 2048));
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[2 * dy]     = (short)cv_round((double)(c0 * 2048));
        ibeta[2 * dy + 1] = (short)cv_round((double)(c1 * 2048));
    }
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    std::vector<int> row0(dw), row1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {                      // HResizeLinear
            int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
            int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)                        // VResizeLinear<uchar,...>
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// ---- A.1.1: FAST-9/16 with cornerScore and 3x3 NMS ------------------------------------
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

// score map: 0 where not a corner, cornerScore otherwise (what FAST_t keeps in buf[])
void fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold,
                    std::vector<uint8_t>& score) {
    score.assign((size_t)w * h, 0);
    int pixel[25];
    for (int k = 0; k < 25; k++) pixel[k] = RING_DY[k % 16] * stride + RING_DX[k % 16];
    for (int y = 3; y < h - 3; y++) {
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* ptr = img + (size_t)y * stride + x;
            int v = ptr[0];
            // contiguous-arc test, 9 of 16 (count > K over the 25-long wrapped ring)
            bool corner = false;
            int vt = v - threshold, cnt = 0;
            for (int k = 0; k < 25; k++) {
                if (ptr[pixel[k]] < vt) { if (++cnt > 8) { corner = true; break; } } else cnt = 0;
            }
            if (!corner) {
                vt = v + threshold; cnt = 0;
                for (int k = 0; k < 25; k++) {
                    if (ptr[pixel[k]] > vt) { if (++cnt > 8) { corner = true; break; } } else cnt = 0;
                }
            }
            if (corner) score[(size_t)y * w + x] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
    }
}

void fast_detect(const uint8_t* img, int w, int h, int stride, int threshold,
                 std::vector<RawKp>& out, std::vector<uint8_t>* score_out) {
    std::vector<uint8_t> sc;
    fast_score_map(img, w, h, stride, threshold, sc);
    out.clear();
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = sc[(size_t)y * w + x];
            if (!s) continue;
            const uint8_t* p = &sc[(size_t)y * w + x];
            if (s > p[-1] && s > p[1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] &&
                s > p[w - 1] && s > p[w] && s > p[w + 1])
                out.push_back(RawKp{x, y, (float)s});
        }
    if (score_out) score_out->swap(sc);
}

// ---- KeyPointsFilter::retainBest (A.1 item 5): keep every kp with response >= n-th best ---
// What OpenCV 3.2 (features2d/keypoint.cpp) executes, for whoever can diff against a real build:
//     if (n_points >= 0 && keypoints.size() > (size_t)n_points) {
//         if (n_points == 0) { keypoints.clear(); return; }
//         std::nth_element(keypoints.begin(), keypoints.begin() + n_points, keypoints.end(), KeypointResponseGreater());
//         float ambiguous_response = keypoints[n_points - 1].response;          // an element of the UNSORTED top n
//         new_end = std::partition(keypoints.begin() + n_points, keypoints.end(), KeypointResponseGreaterThanThreshold(ambiguous_response));   // >=
//         keypoints.resize(new_end - keypoints.begin());
//     }
// keypoints[n-1] after nth_element is some member of the top n, not necessarily the smallest: with ties at the cut the kept set
// (and its order) is libstdc++-defined.  SPEC of this restatement: cut = the true n-th best response, every tie at the cut is
// kept (what OpenCV does whenever keypoints[n-1] carries the cut value, e.g. when the whole top n is tied), canonical order.
static void retain_best(std::vector<RawKp>& k, int n) {
    if (n < 0 || (int)k.size() <= n) return;
    if (n == 0) { k.clear(); return; }
    std::vector<float> r(k.size());
    for (size_t i = 0; i < k.size(); i++) r[i] = k[i].response;
    std::nth_element(r.begin(), r.begin() + (n - 1), r.end(), std::greater<float>());
    float cut = r[n - 1];
    std::vector<RawKp> o;
    for (auto& q : k) if (q.response >= cut) o.push_back(q);
    k.swap(o);
}

// ---- A.1 item 5: HarrisResponses(block 7, k 0.04) ----------------------------------------
static float harris_response(const uint8_t* img, int stride, int x0, int y0) {
    const int blockSize = 7, r = blockSize / 2;
    float scale = 1.f / ((1 << 2) * blockSize * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < blockSize; i++)
        for (int j = 0; j < blockSize; j++) {
            const uint8_t* ptr = img + (size_t)(y0 - r + i) * stride + (x0 - r + j);
            int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-stride + 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[stride - 1]);
            int Iy = (ptr[stride] - ptr[-stride]) * 2 + (ptr[stride - 1] - ptr[-stride - 1]) + (ptr[stride + 1] - ptr[-stride + 1]);
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy;
        }
    const float harris_k = 0.04f;
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

// ---- A.1 item 6: IC angle ---------------------------------------------------------------
void compute_umax(int halfPatch, std::vector<int>& umax) {
    umax.assign(halfPatch + 2, 0);
    int v, v0, vmax = cv_floor_f(halfPatch * std::sqrt(2.f) / 2 + 1);
    int vmin = (int)std::ceil(halfPatch * std::sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(std::sqrt((double)halfPatch * halfPatch - v * v));
    for (v = halfPatch, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

float fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::fabs(x), ay = std::fabs(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static float ic_angle(const uint8_t* img, int stride, int x0, int y0, const std::vector<int>& umax, int half_k) {
    const uint8_t* center = img + (size_t)y0 * stride + x0;
    int m_01 = 0, m_10 = 0;
    for (int u = -half_k; u <= half_k; ++u) m_10 += u * center[u];
    for (int v = 1; v <= half_k; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// ---- A.1 item 8: GaussianBlur 7x7 sigma 2, 8-bit fixed point, BORDER_REFLECT_101 --------
void gaussian_kernel7_q8(int k[7]) {
    // getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256)
    const int n = 7; const double sigma = 2.0;
    double scale2X = -0.5 / (sigma * sigma), sum = 0;
    float cf[7];
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t; sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) { cf[i] = (float)(cf[i] * sum); k[i] = cv_round((double)cf[i] * 256.0); }
}

static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    int k[7]; gaussian_kernel7_q8(k);
    std::vector<int> tmp((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * src[(size_t)y * sstride + reflect101(x + i, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = -3; i <= 3; i++) s += k[i + 3] * tmp[(size_t)reflect101(y + i, h) * w + x];
            int v = (s + (1 << 15)) >> 16;                      // FixedPtCastEx<int,uchar>(16)
            dst[(size_t)y * dstride + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

static const int8_t PATTERN[256 * 4] = {
#include "orb_pattern.inc"
};

static void brief_descriptor(const uint8_t* blurred, int stride, int cx, int cy, float angle_deg, uint8_t* desc) {
    float angle = angle_deg;
    angle *= (float)(M_PI / 180.f);
    double sd, cd;
    sincos_det((double)angle, &sd, &cd);
    float a = (float)cd, b = (float)sd;
    const uint8_t* center = blurred + (size_t)cy * stride + cx;
    const int8_t* pat = PATTERN;
    auto get = [&](int idx) -> int {
        float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        float x = px * a - py * b;
        float y = px * b + py * a;
        int ix = cv_round((double)x), iy = cv_round((double)y);
        return center[iy * stride + ix];
    };
    for (int i = 0; i < 32; i++, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            int t0 = get(2 * k), t1 = get(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

// ---- the whole detectAndCompute ----------------------------------------------------------
int orb_detect_compute(const vis_params& p, const uint8_t* img, int w, int h, int stride,
                       std::vector<vis_keypoint>& kps, std::vector<uint8_t>& desc) {
    if (p.nlevels < 1 || p.nlevels > VIS_MAX_LEVELS || w < 8 || h < 8) return VIS_E_INVALID;
    std::vector<LevelGeom> g;
    level_geometry(p, w, h, g);
    const int L = p.nlevels;
    std::vector<std::vector<uint8_t>> pyr(L);
    pyr[0].resize((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(&pyr[0][(size_t)y * w], img + (size_t)y * stride, w);
    for (int l = 1; l < L; l++) {
        if (g[l].w < 1 || g[l].h < 1) return VIS_E_INVALID;
        pyr[l].resize((size_t)g[l].w * g[l].h);
        resize_linear(pyr[l - 1].data(), g[l - 1].w, g[l - 1].h, g[l - 1].w, pyr[l].data(), g[l].w, g[l].h, g[l].w);
    }
    std::vector<int> umax; compute_umax(p.patch_size / 2, umax);
    const int edge = p.edge_threshold;
    kps.clear(); desc.clear();
    std::vector<std::vector<RawKp>> per_level(L);
    for (int l = 0; l < L; l++) {
        std::vector<RawKp> k;
        fast_detect(pyr[l].data(), g[l].w, g[l].h, g[l].w, p.fast_threshold, k, nullptr);
        // KeyPointsFilter::runByImageBorder(keypoints, img.size(), edgeThreshold)
        std::vector<RawKp> kb;
        for (auto& q : k)
            if (q.x >= edge && q.x < g[l].w - edge && q.y >= edge && q.y < g[l].h - edge) kb.push_back(q);
        retain_best(kb, 2 * g[l].quota);                          // HARRIS_SCORE: keep 2x
        for (auto& q : kb) q.response = harris_response(pyr[l].data(), g[l].w, q.x, q.y);
        retain_best(kb, g[l].quota);
        // canonical order inside a level (OpenCV's order here is a libstdc++ artefact):
        std::sort(kb.begin(), kb.end(), [](const RawKp& A, const RawKp& B) {
            if (A.response != B.response) return A.response > B.response;
            if (A.y != B.y) return A.y < B.y;
            return A.x < B.x;
        });
        per_level[l].swap(kb);
    }
    for (int l = 0; l < L; l++) {
        std::vector<uint8_t> blurred((size_t)g[l].w * g[l].h);
        gaussian_blur7(pyr[l].data(), g[l].w, g[l].h, g[l].w, blurred.data(), g[l].w);
        float sf = g[l].scale;
        for (auto& q : per_level[l]) {
            vis_keypoint kp;
            kp.angle = ic_angle(pyr[l].data(), g[l].w, q.x, q.y, umax, p.patch_size / 2);
            kp.x = (float)q.x * sf; kp.y = (float)q.y * sf;       // pt *= layerScale
            kp.size = p.patch_size * sf;
            kp.response = q.response; kp.octave = l; kp.class_id = -1;
            // computeOrbDescriptors: centre = cvRound(pt * (1.f/layerScale))
            float inv = 1.f / sf;
            int cx = cv_round((double)(kp.x * inv)), cy = cv_round((double)(kp.y * inv));
            uint8_t d[32];
            brief_descriptor(blurred.data(), g[l].w, cx, cy, kp.angle, d);
            kps.push_back(kp);
            desc.insert(desc.end(), d, d + 32);
        }
    }
    return VIS_OK;
}

}  // namespace orc

// ---- C API ---------------------------------------------------------------------------------
using namespace orc;

extern "C" int orc_level_geometry(const vis_params* p, int w, int h, int32_t* widths, int32_t* heights,
                                  float* scales, int32_t* quotas) {
    if (!p || p->nlevels < 1 || p->nlevels > VIS_MAX_LEVELS) return VIS_E_INVALID;
    std::vector<LevelGeom> g; level_geometry(*p, w, h, g);
    for (int l = 0; l < p->nlevels; l++) {
        if (widths) widths[l] = g[l].w;
        if (heights) heights[l] = g[l].h;
        if (scales) scales[l] = g[l].scale;
        if (quotas) quotas[l] = g[l].quota;
    }
    return VIS_OK;
}

extern "C" int orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride,
                                 uint8_t* dst, int dw, int dh, int dstride) {
    if (!src || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1) return VIS_E_INVALID;
    resize_linear(src, sw, sh, sstride, dst, dw, dh, dstride);
    return VIS_OK;
}

// Camera::Update, src/Camera.cpp:68-70: resize(prev, next, Size(), 0.5, 0.5).  What cv::resize (OpenCV 3.2 imgproc/imgwarp.cpp)
// does with these arguments, restated from the published source (PARITY UNPINNED like everything OpenCV-owned here):
//   dsize = Size(saturate_cast<int>(ssize.width * 0.5), saturate_cast<int>(ssize.height * 0.5))   -- cvRound: round half to EVEN
//   scale_x = scale_y = 1 / 0.5 = 2 exactly (inv_scale stays the fx / fy given, because dsize was empty), so is_area_fast holds and
//   "INTER_LINEAR && is_area_fast && iscale == 2" is rerouted to INTER_AREA -> resizeAreaFast_Invoker:
//     destination pixels whose 2x2 source block is complete:  (a + b + c + d + 2) >> 2
//     the others (last column when dsize.width * 2 > ssize.width, last row likewise):
//         saturate_cast<uchar>((float)sum / count) over the source pixels of the block that exist  (round half to even)
// For sizes that halve exactly (752x480 -> 47x30) only the first form occurs.  A size like 137 halves to 68 (68.5 -> even):
// the last source column is then simply unused.
static inline int half_dim(int n) { return (n >> 1) + ((n & 1) & ((n >> 1) & 1)); }
extern "C" void orc_half_pyramid_dims(int w, int h, int32_t lw[5], int32_t lh[5]) {
    lw[0] = w; lh[0] = h;
    for (int l = 1; l < 5; l++) { lw[l] = half_dim(lw[l - 1]); lh[l] = half_dim(lh[l - 1]); }
}
extern "C" int orc_half_pyramid(const uint8_t* img, int w, int h, int stride, uint8_t* const out_levels[5]) {
    if (!img || !out_levels || w < 16 || h < 16) return VIS_E_INVALID;
    std::vector<uint8_t> prev((size_t)w * h);
    for (int y = 0; y < h; y++) std::memcpy(&prev[(size_t)y * w], img + (size_t)y * stride, w);
    if (out_levels[0]) std::memcpy(out_levels[0], prev.data(), prev.size());
    int pw = w, ph = h;
    for (int l = 1; l < 5; l++) {
        const int nw = half_dim(pw), nh = half_dim(ph);
        std::vector<uint8_t> cur((size_t)nw * nh);
        for (int y = 0; y < nh; y++)
            for (int x = 0; x < nw; x++) {
                if (2 * x + 1 < pw && 2 * y + 1 < ph) {
                    const uint8_t* s = &prev[(size_t)(2 * y) * pw + 2 * x];
                    cur[(size_t)y * nw + x] = (uint8_t)((s[0] + s[1] + s[pw] + s[pw + 1] + 2) >> 2);
                } else {
                    int sum = 0, count = 0;
                    for (int sy = 2 * y; sy < 2 * y + 2 && sy < ph; sy++)
                        for (int sx = 2 * x; sx < 2 * x + 2 && sx < pw; sx++) { sum += prev[(size_t)sy * pw + sx]; count++; }
                    cur[(size_t)y * nw + x] = (uint8_t)std::lrintf((float)sum / (float)count);      // count is 1 or 2: exact, <= 255
                }
            }
        if (out_levels[l]) std::memcpy(out_levels[l], cur.data(), cur.size());
        prev.swap(cur); pw = nw; ph = nh;
    }
    return VIS_OK;
}

extern "C" int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold,
                               int32_t* xs, int32_t* ys, int32_t* scores, int cap, uint8_t* score_map) {
    if (!img || w < 7 || h < 7) return VIS_E_INVALID;
    std::vector<RawKp> k; std::vector<uint8_t> sc;
    fast_detect(img, w, h, stride, threshold, k, &sc);
    if (score_map) std::memcpy(score_map, sc.data(), sc.size());
    if ((int)k.size() > cap) return VIS_E_CAPACITY;
    for (size_t i = 0; i < k.size(); i++) {
        if (xs) xs[i] = k[i].x;
        if (ys) ys[i] = k[i].y;
        if (scores) scores[i] = (int)k[i].response;
    }
    return (int)k.size();
}

extern "C" int orc_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
    if (!src || !dst || w < 1 || h < 1) return VIS_E_INVALID;
    gaussian_blur7(src, w, h, sstride, dst, dstride);
    return VIS_OK;
}

extern "C" int orc_orb_detect_compute(const vis_params* p, const uint8_t* img, int w, int h, int stride,
                                      vis_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
    if (!p || !img || !n_out) return VIS_E_INVALID;
    std::vector<vis_keypoint> k; std::vector<uint8_t> d;
    int rc = orb_detect_compute(*p, img, w, h, stride, k, d);
    if (rc) return rc;
    *n_out = (int)k.size();
    if ((int)k.size() > cap) return VIS_E_CAPACITY;
    if (kps && !k.empty()) std::memcpy(kps, k.data(), k.size() * sizeof(vis_keypoint));
    if (desc && !d.empty()) std::memcpy(desc, d.data(), d.size());
    return VIS_OK;
}

// ---- small exports for the unit tests ----
extern "C" void orc_sincos_det(double x, double* s, double* c) { orc::sincos_det(x, s, c); }
extern "C" float orc_fast_atan2(float y, float x) { return orc::fast_atan2(y, x); }
extern "C" void orc_umax(int half_patch, int* out) { std::vector<int> u; orc::compute_umax(half_patch, u); for (int i = 0; i <= half_patch; i++) out[i] = u[i]; }
extern "C" void orc_gaussian_kernel7_q8(int* k) { orc::gaussian_kernel7_q8(k); }
