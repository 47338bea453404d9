// lds_gather.hip -- what does a wave-level gather of 16 bytes per lane from RANDOM 4-byte-aligned LDS addresses cost on gfx950, by
// instruction form?  (k_describe's vertical blur taps: 8 such gathers per lane and keypoint; SQ_LDS_IDX_ACTIVE says its LDS array is
// 97 % busy, 59 % of that bank conflicts.)  Forms: 2 x ds_read2_b32 (what the kernel issues), 2 x ds_read_b64 and 1 x ds_read_b128
// at 4-byte alignment (gfx950 executes them; are they cheaper?), and for reference 1 x ds_read_b128 at 16-byte alignment and
// 1 x ds_read_u16.  Results are checked against a host replay.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_gather.hip -o vi-slam_amd/lib/lds_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define REGION 3696            // bytes per wave: 40 columns x 92 bytes + pad (k_describe's transposed blur buffer)
__host__ __device__ inline uint32_t lcg(uint32_t s) { return s * 1664525u + 1013904223u; }
__host__ __device__ inline uint32_t addr_of(uint32_t s, int form) {
    const uint32_t c = (s >> 8) % 37u, rh = (s >> 20) % 20u;
    uint32_t a = (23u * c + rh) * 4u;
    if (form == 3) a &= ~15u;
    if (form == 4) a += (s >> 4) & 2u;
    return a;
}
template <int FORM>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * REGION];
    for (int i = threadIdx.x; i < 4 * REGION; i += 256) lds[i] = (uint8_t)(i * 7 + (i >> 8));
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds + (threadIdx.x >> 6) * REGION;
    uint32_t s = threadIdx.x * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < iters; it++) {
        s = lcg(s);
        const uint32_t a = base + addr_of(s, FORM);
        uint32_t w0, w1, w2, w3;
        if (FORM == 0) {
            uint64_t p, q;
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p), "=&v"(q) : "v"(a) : "memory");
            w0 = (uint32_t)p; w1 = (uint32_t)(p >> 32); w2 = (uint32_t)q; w3 = (uint32_t)(q >> 32);
        } else if (FORM == 1) {
            uint64_t p, q;
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:8\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p), "=&v"(q) : "v"(a) : "memory");
            w0 = (uint32_t)p; w1 = (uint32_t)(p >> 32); w2 = (uint32_t)q; w3 = (uint32_t)(q >> 32);
        } else if (FORM == 2 || FORM == 3) {
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 p;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(p) : "v"(a) : "memory");
            w0 = p.x; w1 = p.y; w2 = p.z; w3 = p.w;
        } else {
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w0) : "v"(a) : "memory");
            w1 = w2 = w3 = 0;
        }
        acc += w0 + w1 * 3u + w2 * 5u + w3 * 7u;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
static uint32_t host_ref(int tid, int iters, int form) {
    std::vector<uint8_t> lds(4 * REGION);
    for (int i = 0; i < 4 * REGION; i++) lds[i] = (uint8_t)(i * 7 + (i >> 8));
    uint32_t s = tid * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < iters; it++) {
        s = lcg(s);
        const uint32_t a = (tid >> 6) * REGION + addr_of(s, form);
        uint32_t w[4] = {0, 0, 0, 0};
        const int nb = form == 4 ? 2 : 16;
        for (int j = 0; j < nb; j++) w[j >> 2] |= (uint32_t)lds[a + j] << (8 * (j & 3));
        acc += w[0] + w[1] * 3u + w[2] * 5u + w[3] * 7u;
    }
    return acc;
}
template <int FORM> static void run(uint32_t* d, const char* name) {
    const int blocks = 256 * 7 * 4, iters = 2000;
    hipLaunchKernelGGL(k<FORM>, dim3(8), dim3(256), 0, 0, d, 64);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) if (h[t] != host_ref(t, 64, FORM)) bad++;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s mismatches %3d / 256   %.3f ms   %.1f ns per wave-level gather per CU\n", name, bad, ms, ms * 1e6 * 256 / ((double)blocks * 4 * iters));
}
int main() {
    uint32_t* d; hipMalloc(&d, (size_t)256 * 7 * 4 * 256 * 4);
    run<0>(d, "16 B: 2 x ds_read2_b32, 4-byte aligned");
    run<1>(d, "16 B: 2 x ds_read_b64, 4-byte aligned");
    run<2>(d, "16 B: 1 x ds_read_b128, 4-byte aligned");
    run<3>(d, "16 B: 1 x ds_read_b128, 16-byte aligned");
    run<4>(d, " 2 B: 1 x ds_read_u16");
    hipFree(d);
    return 0;
}
