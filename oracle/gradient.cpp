/*
 * gradient.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, see vis_oracle.h) for the step that follows matching
 * inside CameraGPU::addGPUKeyframe (/root/reference/src/CameraGPU.cpp:154-157):
 *   Camera::computeGradient                   /root/reference/src/Camera.cpp:167-184
 *   Camera::ObtainPatchesPointsPreviousFrame  /root/reference/src/Camera.cpp:358-410
 *   Camera::ObtainDebugPointsPreviousFrame    /root/reference/src/Camera.cpp:413-445
 *
 * PARITY UNPINNED versus real OpenCV: cv::Scharr / convertScaleAbs / addWeighted are restated from their
 * published definitions (OpenCV 3.x imgproc/core), the reference holds no fixture for them.
 *   cv::Scharr(src, dst, ddepth, dx, dy, scale = 1, delta = 0, borderType): correlation with
 *       [-3 0 3; -10 0 10; -3 0 3] (dx) / its transpose (dy), BORDER_REFLECT_101, times `scale`.  Scharr has no
 *       ksize argument, so the reference's call Scharr(img, gx, CV_16S, 1, 0, 3, 0, BORDER_DEFAULT) means
 *       scale = 3, delta = 0: every value is 3x the plain Scharr response (|v| <= 3*16*255 = 12240, exact in int16).
 *   convertScaleAbs(v, 1, 0): saturate_cast<uchar>(|v|) = min(|v|, 255)
 *   addWeighted(a, 0.5, b, 0.5, 0): saturate_cast<uchar>(a*0.5f + b*0.5f) -- cvRound, i.e. round half to even
 */
#include "vis_oracle.h"
#include <cmath>
#include <cstdlib>

static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

extern "C" int orc_scharr_gradient(const uint8_t* img, int w, int h, int stride, int scale,
                                   int16_t* gx, int16_t* gy, uint8_t* g) {
    if (!img || w < 1 || h < 1 || stride < w) return VIS_E_INVALID;
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = img + (size_t)reflect101(y - 1, h) * stride;
        const uint8_t* r1 = img + (size_t)y * stride;
        const uint8_t* r2 = img + (size_t)reflect101(y + 1, h) * stride;
        for (int x = 0; x < w; x++) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            int vx = 3 * (r0[xp] - r0[xm]) + 10 * (r1[xp] - r1[xm]) + 3 * (r2[xp] - r2[xm]);
            int vy = 3 * (r2[xm] - r0[xm]) + 10 * (r2[x] - r0[x]) + 3 * (r2[xp] - r0[xp]);
            vx *= scale; vy *= scale;
            // saturate_cast<short>
            vx = vx < -32768 ? -32768 : (vx > 32767 ? 32767 : vx);
            vy = vy < -32768 ? -32768 : (vy > 32767 ? 32767 : vy);
            if (gx) gx[(size_t)y * w + x] = (int16_t)vx;
            if (gy) gy[(size_t)y * w + x] = (int16_t)vy;
            if (g) {
                const int a = std::min(std::abs(vx), 255), b = std::min(std::abs(vy), 255);
                const float f = (float)a * 0.5f + (float)b * 0.5f;         // exact: halves of small integers
                g[(size_t)y * w + x] = (uint8_t)std::lrintf(f);           // cvRound: round half to even, <= 255
            }
        }
    }
    return VIS_OK;
}

// src/Camera.cpp:358-410.  `patch_size[lvl] - 1 / 2` is integer arithmetic: 1/2 == 0, so start_point == patch_size.
extern "C" int orc_patch_points(const vis_keypoint* good, int n, const int32_t* lw, const int32_t* lh, int level,
                                float* xyzw, int cap, int* n_out) {
    if (!good || !lw || !lh || !n_out || level < 0 || level > 4) return VIS_E_INVALID;
    static const int patch_size[5] = {5, 3, 2, 5, 5};
    const float factor_lvl = (float)(1.0 / std::pow(2, level));
    const int start_point = patch_size[level] - 1 / 2;
    int cnt = 0;
    const int m = n < 200 ? n : 200;
    for (int k = 0; k < m; k++) {
        const float x = (float)(((double)good[k].x + 0.5) * (double)factor_lvl - 0.5);
        const float y = (float)(((double)good[k].y + 0.5) * (double)factor_lvl - 0.5);
        for (int i = (int)(x - (float)start_point); (float)i <= x + (float)start_point; i++)
            for (int j = (int)(y - (float)start_point); (float)j <= y + (float)start_point; j++)
                if (i > 0 && i < lw[level] && j > 0 && j < lh[level]) {
                    if (xyzw && cnt < cap) { xyzw[4 * cnt] = (float)i; xyzw[4 * cnt + 1] = (float)j; xyzw[4 * cnt + 2] = 1.0f; xyzw[4 * cnt + 3] = 1.0f; }
                    cnt++;
                }
    }
    *n_out = cnt;
    return (xyzw && cnt > cap) ? VIS_E_CAPACITY : VIS_OK;
}

// src/Camera.cpp:413-445
extern "C" int orc_debug_points(const vis_keypoint* good, int n, int level, float* xyzw, int cap, int* n_out) {
    if (!good || !n_out || level < 0 || level > 4) return VIS_E_INVALID;
    const float factor_lvl = (float)(1.0 / std::pow(2, level));
    const int m = n < 200 ? n : 200;
    for (int k = 0; k < m && k < cap; k++) {
        xyzw[4 * k] = (float)(((double)good[k].x + 0.5) * (double)factor_lvl - 0.5);
        xyzw[4 * k + 1] = (float)(((double)good[k].y + 0.5) * (double)factor_lvl - 0.5);
        xyzw[4 * k + 2] = 1.0f; xyzw[4 * k + 3] = 1.0f;
    }
    *n_out = m;
    return (xyzw && m > cap) ? VIS_E_CAPACITY : VIS_OK;
}
