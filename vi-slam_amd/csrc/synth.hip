// synth.hip -- device-side generator of the synthetic streams: the same integer rule as the host
// generator (synth_core.h), one thread per 4 pixels, frames written straight into HBM.  Test / bench tooling of the
// library (the reference has no data generator; its input is the EuRoC dataset, which is not in this environment):
// it removes the host-side frame synthesis (0.7 G pixel hashes for a 2048-frame ring) and the H2D copy from the start-up
// of every rank of a multi-GPU run.
#include "vis_internal.h"
#include "synth_core.h"
#include <vector>

__global__ __launch_bounds__(256) void k_synth(const uint8_t* __restrict__ canvas, int dim, unsigned long long seed, int t0, int n,
                                               int w, int h, int stride, int mode, const SynthOrigin* __restrict__ org,
                                               uint8_t* __restrict__ out) {
    const int f = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int w4 = (w + 3) / 4;
    if (f >= n || idx >= w4 * h) return;
    const int y = idx / w4, x4 = (idx - y * w4) * 4;
    const SynthOrigin o = org[f];
    uint8_t* d = out + ((size_t)f * h + y) * stride + x4;
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (x4 + k < w) v |= (uint32_t)synth_pixel(canvas, dim, seed, t0 + f, w, x4 + k, y, mode, o) << (8 * k);
    if (x4 + 3 < w && (stride & 3) == 0) *reinterpret_cast<uint32_t*>(d) = v;
    else for (int k = 0; k < 4 && x4 + k < w; k++) d[k] = (uint8_t)(v >> (8 * k));
}

extern "C" int vis_synth_frames_device(vis_ctx* ctx, const uint8_t* d_canvas, int dim, uint64_t seed, int t0, int n,
                                       int w, int h, int stride, int mode, uint8_t* d_out) {
    if (!ctx || !d_canvas || !d_out || n < 1 || stride < w || (mode != 0 && mode != 1) || t0 < 0) return VIS_E_INVALID;
    (void)hipSetDevice(ctx->device);
    std::vector<SynthOrigin> org((size_t)n);
    for (int i = 0; i < n; i++) { const int rc = vis_synth_origin(dim, seed, t0 + i, w, h, &org[(size_t)i]); if (rc) return rc; }
    int rc = vis_ensure_scratch(ctx, (size_t)n * sizeof(SynthOrigin) + 1024);
    if (rc) return rc;
    SynthOrigin* d_org = reinterpret_cast<SynthOrigin*>(ctx->d_scratch);
    HIPCHK(ctx, hipMemcpyAsync(d_org, org.data(), (size_t)n * sizeof(SynthOrigin), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));                  // `org` leaves scope; pageable source
    const int items = ((w + 3) / 4) * h;
    hipLaunchKernelGGL(k_synth, dim3((items + 255) / 256, n), dim3(256), 0, ctx->stream, d_canvas, dim, (unsigned long long)seed, t0, n,
                       w, h, stride, mode, d_org, d_out);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}
