// lds_unaligned.hip -- are 2-byte-aligned ds_read_b128 / ds_read_b64 / ds_read_b32 exact on gfx950, and what do they cost?
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned.hip -o vi-slam_amd/lib/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
struct __attribute__((packed, aligned(2))) Q4 { uint32_t a, b, c, d; };
struct __attribute__((packed, aligned(1))) Q4b { uint32_t a, b, c, d; };

template <int ALIGN_MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, int salt) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (uint8_t)((i * 7 + (i >> 8)) ^ salt);
    __syncthreads();
    uint32_t acc = 0;
    int o = (threadIdx.x * 92 + (ALIGN_MODE == 0 ? 0 : ALIGN_MODE == 1 ? 2 * (threadIdx.x & 1) : (threadIdx.x & 3))) & 8191;
    for (int it = 0; it < iters; it++) {
        if (ALIGN_MODE == 2) { const Q4b q = *reinterpret_cast<const Q4b*>(lds + o); acc += q.a + q.b * 3 + q.c * 5 + q.d * 7; }
        else { const Q4 q = *reinterpret_cast<const Q4*>(lds + o); acc += q.a + q.b * 3 + q.c * 5 + q.d * 7; }
        o = (o + 184) & 8191;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static uint32_t host_ref(int tid, int iters, int salt, int mode) {
    std::vector<uint8_t> lds(16384);
    for (int i = 0; i < 16384; i++) lds[i] = (uint8_t)((i * 7 + (i >> 8)) ^ salt);
    uint32_t acc = 0;
    int o = (tid * 92 + (mode == 0 ? 0 : mode == 1 ? 2 * (tid & 1) : (tid & 3))) & 8191;
    for (int it = 0; it < iters; it++) {
        uint32_t w[4];
        for (int j = 0; j < 4; j++) w[j] = lds[o + 4 * j] | (lds[o + 4 * j + 1] << 8) | (lds[o + 4 * j + 2] << 16) | ((uint32_t)lds[o + 4 * j + 3] << 24);
        acc += w[0] + w[1] * 3 + w[2] * 5 + w[3] * 7;
        o = (o + 184) & 8191;
    }
    return acc;
}

template <int MODE> static void run(uint32_t* d, const char* name) {
    const int blocks = 2048, iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 5);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) if (h[t] != host_ref(t, 16, 5, MODE)) bad++;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s mismatches %d / 256   %.3f ms  (%.1f cycles per wave-level 16-byte read per CU)\n", name, bad, ms,
           ms * 1e-3 * 2.4e9 * 256 / ((double)blocks * 4 * iters));
}

int main() {
    uint32_t* d; hipMalloc(&d, 2048 * 256 * 4);
    run<0>(d, "16-byte reads, 4-byte aligned");
    run<1>(d, "16-byte reads, half of them 2 mod 4");
    run<2>(d, "16-byte reads, byte aligned (any)");
    hipFree(d);
    return 0;
}
