// mfma_valu_overlap.hip -- do v_mfma_i32_32x32x32_i8 chains and plain VALU work overlap on one gfx950 SIMD?
// modes: MFMA only | VALU only | both in every wave (interleaved by the compiler's order) | half the waves each
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o vi-slam_amd/lib/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    const int wave = threadIdx.x >> 6;
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    v16i acc0 = {0}, acc1 = {0};
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 3 + i;
    const uint32_t y = blockIdx.x + 11;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < iters; it++) {
        if (do_m) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
            }
        }
        if (do_v) {
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> static float run(uint32_t* d, const char* name, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu, iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %d waves/SIMD  %8.3f ms\n", name, wgs_per_cu, ms);
    return ms;
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {2, 8}) {
        // per iteration and wave: 8 MFMA (8 x 32 = 256 cycles of the matrix pipe) and 48 VALU (48 x 4.2 = 200 cycles)
        const float m = run<0>(d, "MFMA only (8 per iteration)", w);
        const float v = run<1>(d, "VALU only (48 v_pk_min_i16 per iteration)", w);
        const float b = run<2>(d, "both, in every wave", w);
        const float h = run<3>(d, "even waves MFMA, odd waves VALU (half the work)", w);
        printf("  -> both / (MFMA + VALU) = %.2f, both / max = %.2f;  split: %.3f vs max(m, v) / 2 = %.3f, (m + v) / 2 = %.3f\n\n",
               b / (m + v), b / (m > v ? m : v), h, (m > v ? m : v) / 2, (m + v) / 2);
    }
    hipFree(d);
    return 0;
}
