// Do the FP4 MFMA chains of k_knn_mfma and its top-2 key maintenance (v_med3_i32 + v_min_i32) overlap on a gfx950 SIMD?
// Per iteration and wave: 8 x v_mfma_scale_f32_32x32x64_f8f6f4 (two chains of four) and 32 top-2 updates per chain.
// modes: 0 MFMA only | 1 top-2 only | 2 top-2 on the accumulators the chains just produced (the kernel's order)
//        3 software pipelined: chains of iteration i + 1 issued before the top-2 of iteration i | 4 as 3 with max/min/min (2-source ops)
// hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_top2_overlap.hip -o vi-slam_amd/lib/mfma_top2_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int imed3(int a, int b, int c) { int d; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    v8i a = {(int)threadIdx.x | 0x22222222, 0x2A2A2A2A, 0x22A2A2A2, 0x2222AAAA, 0, 0, 0, 0}, b = {0x2A22A22A, 0x22222222, (int)blockIdx.x | 0x22222222, 0x2AAA2AAA, 0, 0, 0, 0};
    v16f c0, c1, p0, p1;
    for (int i = 0; i < 16; i++) { c0[i] = 1048576.f + i; c1[i] = 1048576.f + i; p0[i] = (float)(threadIdx.x * 16 + i); p1[i] = (float)(threadIdx.x * 7 + i); }
    int k0 = 0x7FFFFFFF, k1 = 0x7FFFFFFF, j0 = 0x7FFFFFFF, j1 = 0x7FFFFFFF;
    for (int it = 0; it < iters; it++) {
        if (MODE != 1) {        // the chains run on (every iteration adds onto the same accumulators: nothing is dead)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 133, 0, 133);
                c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 4, 4, 0, 133, 0, 133);
            }
        }
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                int x = __float_as_int(MODE == 2 ? c0[r] : p0[r]), y = __float_as_int(MODE == 2 ? c1[r] : p1[r]);
                if (MODE == 4) {
                    int t, u;
                    asm volatile("v_max_i32 %0, %1, %2" : "=v"(t) : "v"(k0), "v"(x)); asm volatile("v_min_i32 %0, %0, %1" : "+v"(k0) : "v"(x)); asm volatile("v_min_i32 %0, %0, %1" : "+v"(k1) : "v"(t));
                    asm volatile("v_max_i32 %0, %1, %2" : "=v"(u) : "v"(j0), "v"(y)); asm volatile("v_min_i32 %0, %0, %1" : "+v"(j0) : "v"(y)); asm volatile("v_min_i32 %0, %0, %1" : "+v"(j1) : "v"(u));
                } else {
                    asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(k1) : "v"(k0), "v"(x)); asm volatile("v_min_i32 %0, %0, %1" : "+v"(k0) : "v"(x));
                    asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(j1) : "v"(j0), "v"(y)); asm volatile("v_min_i32 %0, %0, %1" : "+v"(j0) : "v"(y));
                }
            }
        }
        if (MODE >= 3) { p0 = c0; p1 = c1; }      // the next iteration's top-2 works on a copy while the chains run on
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + p0[i] + p1[i];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s + k0 + k1 + j0 + j1;
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
    printf("%-64s %d waves/SIMD  %8.3f ms  %.0f cycles per iteration per SIMD\n", name, wgs_per_cu, ms, ms * 1e-3 * 2.4e9 / (iters * wgs_per_cu));
    return ms;
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 4}) {
        run<0>(d, "8 FP4 MFMA only", w);
        run<1>(d, "32 top-2 updates (med3 + min) only", w);
        run<2>(d, "MFMA, then top-2 on their results (kernel order)", w);
        run<3>(d, "MFMA of the next tile issued before the top-2 of this one", w);
        run<4>(d, "as above with max / min / min (two-source ops)", w);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
