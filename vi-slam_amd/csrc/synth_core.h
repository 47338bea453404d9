// synth_core.h -- the per-pixel rule of the synthetic EuRoC-shaped streams, shared by the host
// generator (geometry.cpp) and the device generator (synth.hip) so that both produce identical bytes.  Integer only.
//   mode 0  "S-752":  crop of one static canvas moving (12, 8) px per frame + noise in {-2..2}.  Planar, pure image
//                     translation: every match is an exact inlier of one essential matrix (best case for RANSAC).
//   mode 1  "S-752P": the same background plus a nearer layer (64 x 64 blocks, 22 % coverage) moving 1.5 x as fast in
//                     the same direction (parallax of a second depth under the same camera translation) and a layer of
//                     independently moving objects (48 x 48 blocks, 6 %, moving (-7, +15) px per frame) that no single
//                     essential matrix explains: outliers + occlusion boundaries, so adaptive RANSAC has work to do.
#ifndef VIS_SYNTH_CORE_H_
#define VIS_SYNTH_CORE_H_
#include <stdint.h>

#if defined(__HIPCC__)
#define SYNTH_FN __host__ __device__ inline
#else
#define SYNTH_FN inline
#endif

struct SynthOrigin { int x0, y0, mx, my, ux, uy; };

SYNTH_FN uint64_t synth_mix(uint64_t z) {                  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

SYNTH_FN uint8_t synth_pixel(const uint8_t* canvas, int dim, uint64_t seed, int t, int w, int x, int y, int mode, SynthOrigin o) {
    int base = canvas[(size_t)(o.y0 + y) * dim + (o.x0 + x)];
    if (mode == 1) {
        const int u = x + o.mx, v = y + o.my;                                      // nearer layer
        const uint64_t hm = synth_mix(seed ^ 0x6A09E667F3BCC909ULL ^ ((uint64_t)(uint32_t)(u >> 6) << 32) ^ (uint64_t)(uint32_t)(v >> 6));
        if (hm % 100 < 22) base = canvas[(size_t)((v + 977) % dim) * dim + ((u + 2311) % dim)];
        const int p = x + o.ux, q = y + o.uy;                                      // independently moving objects
        const uint64_t ho = synth_mix(seed ^ 0xBB67AE8584CAA73BULL ^ ((uint64_t)(uint32_t)(p / 48) << 32) ^ (uint64_t)(uint32_t)(q / 48));
        if (ho % 100 < 6) base = canvas[(size_t)((q + 3001) % dim) * dim + ((p + 613) % dim)];
    }
    // counter-based per-pixel noise in {-2..2}
    const uint64_t z = synth_mix(seed + 0x9E3779B97F4A7C15ULL * (((uint64_t)(uint32_t)t << 32) + (uint64_t)((uint32_t)y * (uint32_t)w + (uint32_t)x) + 1ULL));
    const int val = base + (int)(z % 5) - 2;
    return (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
}
#endif
