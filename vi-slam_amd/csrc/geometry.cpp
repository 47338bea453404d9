// geometry.cpp -- host-side tables of the detector: pyramid level geometry, per-level feature
// quotas, grid-filter band limits, and the
// integer-only synthetic stream generator.  Plain C++ (no device code).
//
// Follows what cv::ORB / cv::resize derive for the reference call site
// /root/reference/src/Camera.cpp:87 (cv::ORB::detectAndCompute) -- see SURVEY.md Appendix A.1
// items 2-3 -- and Matcher::bestMatchesFilter's window arithmetic (/root/reference/src/Matcher.cpp:171-216).
#include "vis_internal.h"
#include "synth_core.h"
#include <cmath>
#include <cstring>

static inline int round_half_even(double v) { return (int)std::lrint(v); }

int vis_compute_levels(const vis_params& p, int w, int h, int stride0, LevelInfo* lv, int fs_nch) {
    if (fs_nch < 1 || fs_nch > VIS_FS_NCH) return VIS_E_INVALID;
    const int emit_h = 8 * fs_nch - 2;                 // emitting rows of a k_fast segment (a plan's choice: see plan_create)
    if (p.nlevels < 1 || p.nlevels > VIS_MAX_LEVELS || p.nfeatures < 1) return VIS_E_INVALID;
    if (w < 2 * p.edge_threshold + 8 || h < 2 * p.edge_threshold + 8 || w > 4095 || h > 4095) return VIS_E_INVALID;
    const double sf = (double)p.scale_factor;            // ORB keeps the float argument in a double member
    for (int l = 0; l < p.nlevels; l++) {
        float s = (float)std::pow(sf, (double)l);
        lv[l].scale = s;
        lv[l].w = round_half_even((double)((float)w / s));
        lv[l].h = round_half_even((double)((float)h / s));
        if (lv[l].w < 8 || lv[l].h < 8) return VIS_E_INVALID;
        lv[l].stride = (l == 0) ? stride0 : ((lv[l].w + 63) / 64) * 64;
        lv[l].frame_bytes = (size_t)lv[l].stride * lv[l].h;
        // FAST items (k_fast, detect.hip): a strip segment of VIS_FS_EMIT_W x (8 fs_nch - 2) emitting positions inside the region that can
        // emit keypoints, [edge, w-edge) x [edge, h-edge); the pixel window of the first strip starts at a dword-aligned column
        // (edge - 4 rounded down to 4; its first emitting column is 4 pixels further).  tiles_x = strips, tiles_y = segments.
        const int e = p.edge_threshold, ex0 = ((e - 4) & ~3) + 4;
        lv[l].tiles_x = std::max(1, (lv[l].w - e - ex0 + VIS_FS_EMIT_W - 1) / VIS_FS_EMIT_W);
        lv[l].tiles_y = std::max(1, (lv[l].h - 2 * e + emit_h - 1) / emit_h);
        lv[l].cand_cap = lv[l].tiles_x * lv[l].tiles_y * VIS_TILE_CAND_CAP;
        lv[l].tile_base = l == 0 ? 0 : lv[l - 1].tile_base + lv[l - 1].tiles_x * lv[l - 1].tiles_y;
    }
    float factor = (float)(1.0 / sf);
    float nd = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)p.nlevels));
    int sum = 0;
    for (int l = 0; l < p.nlevels - 1; l++) {
        lv[l].quota = round_half_even((double)nd);
        sum += lv[l].quota;
        nd *= factor;
    }
    lv[p.nlevels - 1].quota = std::max(p.nfeatures - sum, 0);
    // vis_params.keypoint_capacity beyond the default total is slack that ANY level may use (ties at the Harris cut sit on one level)
    int def_total = 0;
    for (int l = 0; l < p.nlevels; l++) def_total += lv[l].quota + lv[l].quota / 8 + 32;
    const int extra = std::max(0, p.keypoint_capacity - def_total);
    for (int l = 0; l < p.nlevels; l++) {
        int q = lv[l].quota;
        // survivors of the FAST-score cut: 2*quota plus ties at the cut (FAST scores are small
        // integers, ties are common); kept after the Harris cut: quota plus ties.
        int sc = 2 * q + q / 2 + 256;
        int pw = 64; while (pw < sc) pw <<= 1;
        if (pw > 8192) return VIS_E_INVALID;             // LDS sort limit (64 KiB of keys)
        while (pw < sc + extra && pw < 8192) pw <<= 1;    // a raised capacity also widens the one-round sort (k_select's LDS), up to that limit
        lv[l].surv_cap = pw;
        lv[l].keep_cap = std::min(q + q / 8 + 32 + extra, pw);
    }
    return VIS_OK;
}

// Matcher::bestMatchesFilter window limits: winW = w_size/floor(sqrt(n)) stored as float, limits
// accumulated in float exactly like `h_final = h_final + winHSize` (src/Matcher.cpp:177-178,203,229).
void vis_grid_limits(const vis_params& p, int* root, std::vector<float>& hf, std::vector<float>& wf) {
    int r = (int)std::floor(std::sqrt((double)p.n_cells));
    if (r < 1) r = 1;
    if (r > VIS_MAX_GRID_ROOT) r = VIS_MAX_GRID_ROOT;
    *root = r;
    float winW = (float)(p.w_size / std::floor(std::sqrt((double)p.n_cells)));
    float winH = (float)(p.h_size / std::floor(std::sqrt((double)p.n_cells)));
    hf.resize(r); wf.resize(r);
    float a = winH, b = winW;
    for (int j = 0; j < r; j++) { hf[j] = a; wf[j] = b; a = a + winH; b = b + winW; }
}

// ------------------------------------------------------------------------------------------------
// Synthetic EuRoC-shaped stream ("S-752"); integer arithmetic only so the
// bytes are identical on every host.  PRNG = xorshift64*.
struct XorShift64s {
    uint64_t s;
    explicit XorShift64s(uint64_t seed) : s(seed ? seed : 0x9E3779B97F4A7C15ULL) {}
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1DULL; }
    uint32_t below(uint32_t n) { return (uint32_t)((next() >> 33) % n); }
};

extern "C" int vis_synth_canvas(uint8_t* canvas, int dim, uint64_t seed) {
    if (!canvas || dim < 64) return VIS_E_INVALID;
    XorShift64s rng(seed);
    const size_t n = (size_t)dim * dim;
    std::memset(canvas, 128, n);
    const double area_ratio = (double)n / (4096.0 * 4096.0);
    const int nrect = std::max(8, (int)(24000 * area_ratio));
    const int nblob = std::max(8, (int)(12000 * area_ratio));
    for (int i = 0; i < nrect; i++) {
        int rw = 6 + (int)rng.below(59), rh = 6 + (int)rng.below(59);      // 6..64
        int x = (int)rng.below((uint32_t)dim), y = (int)rng.below((uint32_t)dim);
        uint8_t v = (uint8_t)rng.below(256);
        for (int yy = y; yy < std::min(dim, y + rh); yy++) std::memset(canvas + (size_t)yy * dim + x, v, (size_t)std::min(rw, dim - x));
    }
    for (int i = 0; i < nblob; i++) {
        int s = 3 + (int)rng.below(5);                                    // 3..7
        int x = (int)rng.below((uint32_t)dim), y = (int)rng.below((uint32_t)dim);
        uint8_t v = (uint8_t)rng.below(256);
        for (int yy = y; yy < std::min(dim, y + s); yy++) std::memset(canvas + (size_t)yy * dim + x, v, (size_t)std::min(s, dim - x));
    }
    // one 3x3 box blur, replicate border, rounded integer mean
    std::vector<uint8_t> src(canvas, canvas + n);
    for (int y = 0; y < dim; y++) {
        int y0 = y > 0 ? y - 1 : 0, y1 = y < dim - 1 ? y + 1 : dim - 1;
        for (int x = 0; x < dim; x++) {
            int x0 = x > 0 ? x - 1 : 0, x1 = x < dim - 1 ? x + 1 : dim - 1;
            int s = src[(size_t)y0 * dim + x0] + src[(size_t)y0 * dim + x] + src[(size_t)y0 * dim + x1] +
                    src[(size_t)y * dim + x0] + src[(size_t)y * dim + x] + src[(size_t)y * dim + x1] +
                    src[(size_t)y1 * dim + x0] + src[(size_t)y1 * dim + x] + src[(size_t)y1 * dim + x1];
            canvas[(size_t)y * dim + x] = (uint8_t)((s + 4) / 9);
        }
    }
    return VIS_OK;
}

// origins of the layers at frame t (host side: the PRNG draws happen once per stream)
int vis_synth_origin(int dim, uint64_t seed, int t, int w, int h, SynthOrigin* o) {
    if (w < 1 || h < 1 || w >= dim || h >= dim || t < 0 || dim < 64) return VIS_E_INVALID;
    XorShift64s rng(seed ^ 0xD1B54A32D192ED03ULL);
    const int rx = dim - w, ry = dim - h;
    const int ox = (int)rng.below((uint32_t)rx), oy = (int)rng.below((uint32_t)ry);
    o->x0 = (int)(((int64_t)ox + 12LL * t) % rx); o->y0 = (int)(((int64_t)oy + 8LL * t) % ry);
    // the layer offsets stay non-negative for every t < 2^20 so that >> 6 and / 48 are plain floor divisions
    const int mx0 = (int)rng.below(4096), my0 = (int)rng.below(4096), ux0 = (int)rng.below(4096), uy0 = (int)rng.below(4096);
    o->mx = mx0 + 18 * t; o->my = my0 + 12 * t;
    o->ux = ux0 + 48 * 200000 - 7 * t; o->uy = uy0 + 15 * t;
    return VIS_OK;
}

static int synth_frame_mode(const uint8_t* canvas, int dim, uint64_t seed, int t, int w, int h, uint8_t* out, int out_stride, int mode) {
    if (!canvas || !out || out_stride < w) return VIS_E_INVALID;
    SynthOrigin o;
    const int rc = vis_synth_origin(dim, seed, t, w, h, &o);
    if (rc) return rc;
    for (int y = 0; y < h; y++) {
        uint8_t* d = out + (size_t)y * out_stride;
        for (int x = 0; x < w; x++) d[x] = synth_pixel(canvas, dim, seed, t, w, x, y, mode, o);
    }
    return VIS_OK;
}

extern "C" int vis_synth_frame(const uint8_t* canvas, int dim, uint64_t seed, int t,
                               int w, int h, uint8_t* out, int out_stride) {
    return synth_frame_mode(canvas, dim, seed, t, w, h, out, out_stride, 0);
}

extern "C" int vis_synth_frame_parallax(const uint8_t* canvas, int dim, uint64_t seed, int t,
                                        int w, int h, uint8_t* out, int out_stride) {
    return synth_frame_mode(canvas, dim, seed, t, w, h, out, out_stride, 1);
}
