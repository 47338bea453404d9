// gradient.hip -- the step after matching inside CameraGPU::addGPUKeyframe (/root/reference/src/CameraGPU.cpp:154-157)
// on gfx950, batched over frames:
//   Camera::Update's half pyramid            /root/reference/src/Camera.cpp:63-72     k_half_all (k_half4 per level: unaligned buffers)
//   Camera::computeGradient                  /root/reference/src/Camera.cpp:167-184   k_gradient
//   Camera::ObtainPatchesPointsPreviousFrame /root/reference/src/Camera.cpp:358-410   k_patch_points
//   Camera::ObtainDebugPointsPreviousFrame   /root/reference/src/Camera.cpp:413-445   k_debug_points
// Results are bit-identical to oracle/gradient.cpp (tests/test_gradient_gpu.py).
//
// k_gradient is a pure streaming kernel: 1 byte read and 5 bytes written per pixel (gx, gy int16 + blended u8),
// HBM bound.  Thread = 8 pixels x GR_ROWS rows with a register sliding window over rows; a row is one unaligned
// 16-byte load widened to five v_pk_*_i16 operands; the Scharr responses of a pixel pair are 6 packed instructions.
#include "vis_internal.h"

typedef short pk16 __attribute__((ext_vector_type(2)));
typedef unsigned short upk16 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ pk16 as_pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
static __device__ __forceinline__ uint32_t as_u(pk16 v) { return __builtin_bit_cast(uint32_t, v); }

struct GradLevel { int w, h, items_x, items, base_blocks; size_t off; };     // off = element offset of the level inside a frame
struct GradArgs { GradLevel lv[5]; int blocks_per_frame; size_t frame_elems; };

// XCD-aware work mapping (see detect.hip): all blocks of one frame on one XCD
static __device__ __forceinline__ bool xcd_map(int b, int per_frame, int n, int& frame, int& inner) {
    const int xcd = b & 7, j = b >> 3;
    frame = (j / per_frame) * 8 + xcd;
    inner = j - (j / per_frame) * per_frame;
    return frame < n;
}

// ---- Camera::Update: cv::resize's area-fast path (scale exactly 2), 4 output pixels per thread: (a+b+c+d+2)>>2 over every complete
// 2 x 2 source block; where the level is one larger than half of an odd source size, the last column / row averages the source
// pixels that exist -- saturate_cast<uchar>((float)sum / count), count 1 or 2, round half to even (oracle/orb.cpp orc_half_pyramid)
__global__ __launch_bounds__(256) void k_half4(const uint8_t* __restrict__ src, int sw, int sh, int sstride, size_t sframe,
                                               uint8_t* __restrict__ dst, int dw, int dh, size_t dframe, int nframes) {
    const int gx4 = (dw + 3) / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (f >= nframes || idx >= gx4 * dh) return;
    const int y = idx / gx4, x4 = (idx - y * gx4) * 4;
    const uint8_t* s0 = src + (size_t)f * sframe + (size_t)(2 * y) * sstride + 2 * x4;
    const uint8_t* s1 = s0 + sstride;
    uint8_t* d = dst + (size_t)f * dframe + (size_t)y * dw + x4;
    if (2 * y + 1 >= sh || 2 * (x4 + 3) + 1 >= sw) {                    // a partial block in this group of four (or in the whole last row)
        for (int k = 0; k < 4 && x4 + k < dw; k++) {
            const int sx = 2 * (x4 + k);
            const bool right = sx + 1 < sw, below = 2 * y + 1 < sh;
            if (right && below) d[k] = (uint8_t)((s0[2 * k] + s0[2 * k + 1] + s1[2 * k] + s1[2 * k + 1] + 2) >> 2);
            else {
                const int sum = s0[2 * k] + (right ? s0[2 * k + 1] : 0) + (below ? s1[2 * k] : 0);
                const int count = 1 + (right ? 1 : 0) + (below ? 1 : 0);
                d[k] = (uint8_t)__float2int_rn((float)sum / (float)count);
            }
        }
    } else if (x4 + 4 <= dw) {
        typedef uint32_t __attribute__((aligned(1))) u32u;
        const uint2 a = make_uint2(*reinterpret_cast<const u32u*>(s0), *reinterpret_cast<const u32u*>(s0 + 4));
        const uint2 b = make_uint2(*reinterpret_cast<const u32u*>(s1), *reinterpret_cast<const u32u*>(s1 + 4));
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t wa = k < 2 ? a.x : a.y, wb = k < 2 ? b.x : b.y;
            const int sh = 16 * (k & 1);
            const uint32_t v = (((wa >> sh) & 255u) + ((wa >> (sh + 8)) & 255u) + ((wb >> sh) & 255u) + ((wb >> (sh + 8)) & 255u) + 2u) >> 2;
            out |= v << (8 * k);
        }
        if ((((uintptr_t)d) & 3) == 0) *reinterpret_cast<uint32_t*>(d) = out;
        else { d[0] = (uint8_t)out; d[1] = (uint8_t)(out >> 8); d[2] = (uint8_t)(out >> 16); d[3] = (uint8_t)(out >> 24); }
    } else {
        for (int k = 0; k < 4 && x4 + k < dw; k++)
            d[k] = (uint8_t)((s0[2 * k] + s0[2 * k + 1] + s1[2 * k] + s1[2 * k + 1] + 2) >> 2);
    }
}

// ---- Camera::computeGradient ---------------------------------------------------------------------------------
#define GR_ROWS 16
static __device__ __forceinline__ int reflect101(int i, int n) {      // one reflection is enough for |overshoot| <= 1, n >= 2
    return n == 1 ? 0 : (i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i));
}

// pixels x0-1 .. x0+8 of one row as five (u16, u16) pairs, columns reflected at the image border
static __device__ __forceinline__ void load_row(const uint8_t* __restrict__ row, int x0, int w, bool fast, uint32_t (&U)[5]) {
    if (fast) {                                                      // bytes x0-4 .. x0+11 are inside the row
        typedef uint32_t v4u1 __attribute__((ext_vector_type(4), aligned(1)));
        const v4u1 q = *reinterpret_cast<const v4u1*>(row + x0 - 4);  // one (possibly unaligned) 16-byte load
        const uint32_t b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3];
        U[0] = __builtin_amdgcn_perm(b1, b0, 0x0c040c03u);           // (x0-1, x0)
        U[1] = __builtin_amdgcn_perm(0u, b1, 0x0c020c01u);           // (x0+1, x0+2)
        U[2] = __builtin_amdgcn_perm(b2, b1, 0x0c040c03u);           // (x0+3, x0+4)
        U[3] = __builtin_amdgcn_perm(0u, b2, 0x0c020c01u);           // (x0+5, x0+6)
        U[4] = __builtin_amdgcn_perm(b3, b2, 0x0c040c03u);           // (x0+7, x0+8)
    } else {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            // columns past the last pixel only feed outputs that are never stored: clamp them into the row
            const int xa = reflect101(min(x0 - 1 + 2 * k, w), w), xb = reflect101(min(x0 + 2 * k, w), w);
            U[k] = (uint32_t)row[xa] | ((uint32_t)row[xb] << 16);
        }
    }
}

__global__ __launch_bounds__(256) void k_gradient(GradArgs G, const uint8_t* __restrict__ img0, int stride0, size_t frame0,
                                                  const uint8_t* __restrict__ pyr, int k3, int k10,
                                                  int16_t* __restrict__ gx, int16_t* __restrict__ gy, uint8_t* __restrict__ gout,
                                                  int nframes) {
    int f, inner;
    if (!xcd_map(blockIdx.x, G.blocks_per_frame, nframes, f, inner)) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 5; k++) if (inner >= G.lv[k].base_blocks) l = k;
    const GradLevel L = G.lv[l];
    const int item = (inner - L.base_blocks) * 256 + threadIdx.x;
    if (item >= L.items) return;
    const int ry = item / L.items_x, cx = item - ry * L.items_x;
    const int x0 = cx * 8, y0 = ry * GR_ROWS;
    const int w = L.w, h = L.h;
    // level 0 is the caller's frame (its own stride), levels 1..4 the dense half pyramid
    const uint8_t* img = l == 0 ? img0 + (size_t)f * frame0 : pyr + (size_t)f * G.frame_elems + L.off;
    const int stride = l == 0 ? stride0 : w;
    const bool fast = x0 >= 4 && x0 + 12 <= w;
    const size_t obase = (size_t)f * G.frame_elems + L.off;
    const pk16 K3 = {(short)k3, (short)k3}, K10 = {(short)k10, (short)k10};
    uint32_t Ua[5], Ub[5], Uc[5];                                    // rows y-1, y, y+1
    load_row(img + (size_t)reflect101(y0 - 1, h) * stride, x0, w, fast, Ua);
    load_row(img + (size_t)min(y0, h - 1) * stride, x0, w, fast, Ub);
#pragma unroll
    for (int r = 0; r < GR_ROWS; r++) {
        const int y = y0 + r;
        load_row(img + (size_t)reflect101(min(y, h - 1) + 1, h) * stride, x0, w, fast, Uc);
        // S = k3*(above + below) + k10*centre (vertical smoothing), D = below - above, 10 columns as 5 pairs
        pk16 S[5], D[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const pk16 a = as_pk(Ua[k]), b = as_pk(Ub[k]), c = as_pk(Uc[k]);
            S[k] = (a + c) * K3 + b * K10;
            D[k] = c - a;
        }
        uint32_t vx[4], vy[4], vg[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // gx(x) = S(x+1) - S(x-1): pair k+1 minus pair k.  gy(x) = k3*(D(x-1) + D(x+1)) + k10*D(x)
            const pk16 gxv = S[k + 1] - S[k];
            const pk16 dm = as_pk(__builtin_amdgcn_alignbit(as_u(D[k + 1]), as_u(D[k]), 16));
            const pk16 gyv = (D[k] + D[k + 1]) * K3 + dm * K10;
            vx[k] = as_u(gxv); vy[k] = as_u(gyv);
            const pk16 z = {0, 0}, c255 = {255, 255};
            const pk16 ax = __builtin_elementwise_min(__builtin_elementwise_max(gxv, z - gxv), c255);
            const pk16 ay = __builtin_elementwise_min(__builtin_elementwise_max(gyv, z - gyv), c255);
            // (ax + ay) / 2 rounded half to even: q = s >> 1, + 1 when s is odd and q is odd
            const uint32_t s = as_u(ax + ay);
            const uint32_t q = (s >> 1) & 0x7FFF7FFFu;
            vg[k] = q + (q & s & 0x00010001u);
        }
        if (y < h) {
            const size_t o = obase + (size_t)y * w + x0;
            if (x0 + 8 <= w) {
                // rows of odd-width levels are not 16-byte aligned: dword stores (x0 is a multiple of 8, w may be odd)
                if (((o * 2) & 15) == 0) {
                    // write-once streams: non-temporal stores keep them from displacing the source rows in L2
                    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(v4u{vx[0], vx[1], vx[2], vx[3]}, reinterpret_cast<v4u*>(gx + o));
                    __builtin_nontemporal_store(v4u{vy[0], vy[1], vy[2], vy[3]}, reinterpret_cast<v4u*>(gy + o));
                } else if ((o & 1) == 0) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        reinterpret_cast<uint32_t*>(gx + o)[k] = vx[k]; reinterpret_cast<uint32_t*>(gy + o)[k] = vy[k];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        gx[o + 2 * k] = (int16_t)(vx[k] & 0xFFFF); gx[o + 2 * k + 1] = (int16_t)(vx[k] >> 16);
                        gy[o + 2 * k] = (int16_t)(vy[k] & 0xFFFF); gy[o + 2 * k + 1] = (int16_t)(vy[k] >> 16);
                    }
                }
                const uint32_t g0 = __builtin_amdgcn_perm(vg[1], vg[0], 0x06040200u), g1 = __builtin_amdgcn_perm(vg[3], vg[2], 0x06040200u);
                if ((o & 7) == 0) {
                    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
                    __builtin_nontemporal_store(v2u{g0, g1}, reinterpret_cast<v2u*>(gout + o));
                }
                else {
#pragma unroll
                    for (int k = 0; k < 4; k++) { gout[o + k] = (uint8_t)(g0 >> (8 * k)); gout[o + 4 + k] = (uint8_t)(g1 >> (8 * k)); }
                }
            } else {
                for (int k = 0; k < 8 && x0 + k < w; k++) {
                    const uint32_t a = vx[k >> 1], b = vy[k >> 1], c = vg[k >> 1];
                    gx[o + k] = (int16_t)((k & 1) ? (a >> 16) : (a & 0xFFFF));
                    gy[o + k] = (int16_t)((k & 1) ? (b >> 16) : (b & 0xFFFF));
                    gout[o + k] = (uint8_t)((k & 1) ? (c >> 16) : c);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 5; k++) { Ua[k] = Ub[k]; Ub[k] = Uc[k]; }
    }
}

// ---- patch / debug point lists (tiny: <= 200 keypoints): one block per level, ordered output via a block scan ----
__global__ __launch_bounds__(256) void k_patch_points(const vis_keypoint* __restrict__ good, int n, int lw0, int lh0,
                                                      float* __restrict__ out, int cap, int32_t* __restrict__ n_out) {
    __shared__ int pre[257];
    const int level = blockIdx.x, tid = threadIdx.x;
    const int patch_size = level == 1 ? 3 : (level == 2 ? 2 : 5);
    const int sp = patch_size - 1 / 2;                               // src/Camera.cpp:376: integer 1/2 == 0
    const int lw = lw0 >> level, lh = lh0 >> level;
    const float factor_lvl = (float)(1.0 / (double)(1 << level));
    const int m = min(n, 200);
    float x = 0, y = 0; int cnt = 0;
    if (tid < m) {
        x = (float)(((double)good[tid].x + 0.5) * (double)factor_lvl - 0.5);
        y = (float)(((double)good[tid].y + 0.5) * (double)factor_lvl - 0.5);
        for (int i = (int)(x - (float)sp); (float)i <= x + (float)sp; i++)
            for (int j = (int)(y - (float)sp); (float)j <= y + (float)sp; j++)
                if (i > 0 && i < lw && j > 0 && j < lh) cnt++;
    }
    pre[tid + 1] = cnt;
    if (tid == 0) pre[0] = 0;
    __syncthreads();
    if (tid == 0) for (int k = 1; k <= 256; k++) pre[k] += pre[k - 1];
    __syncthreads();
    if (tid < m) {
        int o = pre[tid];
        float* dst = out + (size_t)level * cap * 4;
        for (int i = (int)(x - (float)sp); (float)i <= x + (float)sp; i++)
            for (int j = (int)(y - (float)sp); (float)j <= y + (float)sp; j++)
                if (i > 0 && i < lw && j > 0 && j < lh) {
                    if (o < cap) { dst[4 * o] = (float)i; dst[4 * o + 1] = (float)j; dst[4 * o + 2] = 1.0f; dst[4 * o + 3] = 1.0f; }
                    o++;
                }
    }
    if (tid == 0) n_out[level] = pre[256];
}

__global__ __launch_bounds__(256) void k_debug_points(const vis_keypoint* __restrict__ good, int n, float* __restrict__ out, int cap,
                                                      int32_t* __restrict__ n_out) {
    const int level = blockIdx.x, tid = threadIdx.x;
    const float factor_lvl = (float)(1.0 / (double)(1 << level));
    const int m = min(n, 200);
    if (tid < m && tid < cap) {
        float* dst = out + ((size_t)level * cap + tid) * 4;
        dst[0] = (float)(((double)good[tid].x + 0.5) * (double)factor_lvl - 0.5);
        dst[1] = (float)(((double)good[tid].y + 0.5) * (double)factor_lvl - 0.5);
        dst[2] = 1.0f; dst[3] = 1.0f;
    }
    if (tid == 0) n_out[level] = m;
}

// ---- host side ----------------------------------------------------------------------------------------------
size_t vis_grad_frame_elems(int w, int h) {
    size_t t = 0;
    int lw[5], lh[5]; vis_half_dims(w, h, lw, lh);
    for (int l = 0; l < 5; l++) t += (size_t)lw[l] * lh[l];
    return (t + 63) & ~(size_t)63;                                   // frames start on 64-element boundaries
}

// ---- Camera::Update, all four half levels in ONE pass: thread = one 16 x 16 block of the frame (16 aligned 16-byte row loads),
// level 1 = 8 x 8, level 2 = 4 x 4, level 3 = 2 x 2, level 4 = 1 pixel of it, every level the exact 2 x 2 box mean
// (a + b + c + d + 2) >> 2 of the level above.  The frame is read once instead of 1.33 times and nothing is re-read from memory.
// Needs w, h multiples of 16 and a 16-byte aligned frame / stride (the batched path guarantees w, h; alignment is checked by the
// launcher, which otherwise falls back to the per-level kernel).
static __device__ __forceinline__ uint32_t box_rows(uint32_t a, uint32_t b) {          // two source dwords (rows y, y+1) -> 2 outputs in 16-bit fields
    const uint32_t M = 0x00FF00FFu;
    return ((((a & M) + ((a >> 8) & M)) + ((b & M) + ((b >> 8) & M)) + 0x00020002u) >> 2) & M;
}
// `h` only places the levels inside a frame record (their heights are vis_half_dims'); the rows worked on are the 16 * byn rows of
// complete 16 x 16 blocks -- when h is not a multiple of 16 (1080 = 16 * 67 + 8) the launcher finishes the rest with k_half4.
struct HalfLevels { size_t o1, o2, o3, o4; int byn; };
__global__ __launch_bounds__(256) void k_half_all(const uint8_t* __restrict__ src, int w, HalfLevels HL, int sstride, size_t sframe,
                                                  uint8_t* __restrict__ dst, size_t dframe, int nframes) {
    const int bxn = w >> 4, byn = HL.byn;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y;
    if (f >= nframes || idx >= bxn * byn) return;
    const int by = idx / bxn, bx = idx - by * bxn;
    const uint8_t* s0 = src + (size_t)f * sframe + (size_t)(16 * by) * sstride + 16 * bx;
    uint8_t* d = dst + (size_t)f * dframe;
    // level 1: rows 2r, 2r+1 -> 8 pixels = four dwords of 2 x 16-bit fields each
    uint32_t L1[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const uint4 a = *reinterpret_cast<const uint4*>(s0 + (size_t)(2 * r) * sstride);
        const uint4 b = *reinterpret_cast<const uint4*>(s0 + (size_t)(2 * r + 1) * sstride);
        L1[r][0] = box_rows(a.x, b.x); L1[r][1] = box_rows(a.y, b.y); L1[r][2] = box_rows(a.z, b.z); L1[r][3] = box_rows(a.w, b.w);
    }
    const size_t o1 = HL.o1, o2 = HL.o2, o3 = HL.o3, o4 = HL.o4;
    {   // store level 1: 8 rows x 8 bytes (fields -> bytes: v_perm picks bytes 0 and 2 of two field dwords)
        const int w1 = w >> 1;
        uint8_t* p = d + o1 + (size_t)(8 * by) * w1 + 8 * bx;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t lo = __builtin_amdgcn_perm(L1[r][1], L1[r][0], 0x06040200u), hi = __builtin_amdgcn_perm(L1[r][3], L1[r][2], 0x06040200u);
            *reinterpret_cast<uint2*>(p + (size_t)r * w1) = make_uint2(lo, hi);
        }
    }
    // level 2 from level 1: horizontal neighbours are the two fields of one dword
    uint32_t L2[4][2];                                                                 // 4 rows x 4 pixels, 2 fields per dword
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const uint32_t a0 = L1[2 * r][2 * c], a1 = L1[2 * r][2 * c + 1], b0 = L1[2 * r + 1][2 * c], b1 = L1[2 * r + 1][2 * c + 1];
            const uint32_t s0_ = (a0 & 0xFFFFu) + (a0 >> 16) + (b0 & 0xFFFFu) + (b0 >> 16) + 2u;
            const uint32_t s1_ = (a1 & 0xFFFFu) + (a1 >> 16) + (b1 & 0xFFFFu) + (b1 >> 16) + 2u;
            L2[r][c] = (s0_ >> 2) | ((s1_ >> 2) << 16);
        }
    {
        const int w2 = w >> 2;
        uint8_t* p = d + o2 + (size_t)(4 * by) * w2 + 4 * bx;
#pragma unroll
        for (int r = 0; r < 4; r++) *reinterpret_cast<uint32_t*>(p + (size_t)r * w2) = __builtin_amdgcn_perm(L2[r][1], L2[r][0], 0x06040200u);
    }
    uint32_t L3[2];                                                                    // 2 rows x 2 pixels
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t a0 = L2[2 * r][0], a1 = L2[2 * r][1], b0 = L2[2 * r + 1][0], b1 = L2[2 * r + 1][1];
        const uint32_t s0_ = (a0 & 0xFFFFu) + (a0 >> 16) + (b0 & 0xFFFFu) + (b0 >> 16) + 2u;
        const uint32_t s1_ = (a1 & 0xFFFFu) + (a1 >> 16) + (b1 & 0xFFFFu) + (b1 >> 16) + 2u;
        L3[r] = (s0_ >> 2) | ((s1_ >> 2) << 16);
    }
    {
        const int w3 = w >> 3;
        uint8_t* p = d + o3 + (size_t)(2 * by) * w3 + 2 * bx;
#pragma unroll
        for (int r = 0; r < 2; r++) *reinterpret_cast<uint16_t*>(p + (size_t)r * w3) = (uint16_t)((L3[r] & 0xFFu) | ((L3[r] >> 8) & 0xFF00u));
    }
    d[o4 + (size_t)by * (w >> 4) + bx] = (uint8_t)(((L3[0] & 0xFFFFu) + (L3[0] >> 16) + (L3[1] & 0xFFFFu) + (L3[1] >> 16) + 2u) >> 2);
}

// d_pyr: n x frame_elems u8; levels 1..4 are written at their dense offsets (the level-0 part is not touched:
// level 0 is the caller's frame)
int launch_half_pyramid_batch(vis_ctx* ctx, const uint8_t* d_frames, int w, int h, int stride, size_t frame_bytes, int n,
                              uint8_t* d_pyr) {
    const size_t fe = vis_grad_frame_elems(w, h);
    int lw[5], lh[5]; vis_half_dims(w, h, lw, lh);
    size_t loff[5]; loff[0] = 0;
    for (int l = 1; l < 5; l++) loff[l] = loff[l - 1] + (size_t)lw[l - 1] * lh[l - 1];
    // rows [0, 16 * byn): all four levels in one pass when the WIDTH halves exactly four times and everything is 16-byte aligned
    // (1920 x 1080: 67 of 67.5 block rows); the rows below, or everything when that does not hold, level by level with partial blocks
    int row0 = 0;
    if (!(w & 15) && h >= 16 && !(stride & 15) && !(frame_bytes & 15) && !((uintptr_t)d_frames & 15) && !((uintptr_t)d_pyr & 15) && !(fe & 15) &&
        !(loff[1] & 7) && !(loff[2] & 3) && !(loff[3] & 1)) {
        HalfLevels HL; HL.o1 = loff[1]; HL.o2 = loff[2]; HL.o3 = loff[3]; HL.o4 = loff[4]; HL.byn = h >> 4;
        const int blocks = (w >> 4) * HL.byn;
        hipLaunchKernelGGL(k_half_all, dim3((blocks + 255) / 256, n), dim3(256), 0, ctx->stream, d_frames, w, HL, stride, frame_bytes, d_pyr, fe, n);
        row0 = 16 * HL.byn;
    }
    if (row0 < h)
        for (int l = 1; l < 5; l++) {
            const int sw = lw[l - 1], sh = lh[l - 1], dw = lw[l], dh = lh[l];
            if (dw < 1 || dh < 1) return VIS_E_INVALID;
            const int sr0 = row0 >> (l - 1), dr0 = row0 >> l;                // first source / destination row still to do (row0 is a multiple of 16)
            if (dr0 >= dh) continue;
            const uint8_t* src = (l == 1 ? d_frames : d_pyr + loff[l - 1]) + (size_t)sr0 * (l == 1 ? stride : sw);
            const int ss = l == 1 ? stride : sw;
            const size_t sf = l == 1 ? frame_bytes : fe;
            const int items = ((dw + 3) / 4) * (dh - dr0);
            hipLaunchKernelGGL(k_half4, dim3((items + 255) / 256, n), dim3(256), 0, ctx->stream, src, sw, sh - sr0, ss, sf,
                               d_pyr + loff[l] + (size_t)dr0 * dw, dw, dh - dr0, fe, n);
        }
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}

int launch_gradient(vis_ctx* ctx, const uint8_t* d_frames, int w, int h, int stride, size_t frame_bytes, int n,
                    const uint8_t* d_pyr, int scale, int16_t* d_gx, int16_t* d_gy, uint8_t* d_g) {
    GradArgs G;
    G.frame_elems = vis_grad_frame_elems(w, h);
    int lw[5], lh[5]; vis_half_dims(w, h, lw, lh);
    size_t off = 0; int blocks = 0;
    for (int l = 0; l < 5; l++) {
        GradLevel& L = G.lv[l];
        L.w = lw[l]; L.h = lh[l]; L.off = off;
        if (L.w < 2 || L.h < 2) return VIS_E_INVALID;
        L.items_x = (L.w + 7) / 8;
        L.items = L.items_x * ((L.h + GR_ROWS - 1) / GR_ROWS);
        L.base_blocks = blocks;
        blocks += (L.items + 255) / 256;
        off += (size_t)L.w * L.h;
    }
    G.blocks_per_frame = blocks;
    const int grid = 8 * ((n + 7) / 8) * blocks;
    hipLaunchKernelGGL(k_gradient, dim3(grid), dim3(256), 0, ctx->stream, G, d_frames, stride, frame_bytes, d_pyr, 3 * scale, 10 * scale,
                       d_gx, d_gy, d_g, n);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}

int launch_patch_points(vis_ctx* ctx, const vis_keypoint* d_good, int n, int w, int h, float* d_patch, float* d_debug, int cap,
                        int32_t* d_counts /* [10]: 5 patch + 5 debug */) {
    hipLaunchKernelGGL(k_patch_points, dim3(5), dim3(256), 0, ctx->stream, d_good, n, w, h, d_patch, cap, d_counts);
    hipLaunchKernelGGL(k_debug_points, dim3(5), dim3(256), 0, ctx->stream, d_good, n, d_debug, cap, d_counts + 5);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}
