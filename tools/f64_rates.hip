// f64_rates.hip -- issue rate of the double-precision VALU instructions the pose kernels are made of (gfx950), by waves per SIMD and by
// the number of independent dependency chains per wave (CH): cycles per wave64 instruction per SIMD at 2.4 GHz nominal.
// build: hipcc --offload-arch=gfx950 -O3 tools/f64_rates.hip -o vi-slam_amd/lib/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct MulF64 { static __device__ __forceinline__ void op(double& a, double b, double c) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b)); } static const char* nm() { return "v_mul_f64"; } };
struct AddF64 { static __device__ __forceinline__ void op(double& a, double b, double c) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b)); } static const char* nm() { return "v_add_f64"; } };
struct FmaF64 { static __device__ __forceinline__ void op(double& a, double b, double c) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); } static const char* nm() { return "v_fma_f64"; } };
struct MulAdd { static __device__ __forceinline__ void op(double& a, double b, double c) { asm volatile("v_mul_f64 %0, %0, %1\n\tv_add_f64 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c)); } static const char* nm() { return "v_mul_f64 + v_add_f64 (Horner step, 2 instr)"; } };
struct FmaF32 { static __device__ __forceinline__ void op(double& a, double b, double c) { float x = (float)a; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"((float)b), "v"((float)c)); a = x; } static const char* nm() { return "v_fma_f32 (+2 cvt)"; } };

template <class O, int CH>
__global__ __launch_bounds__(256) void k_spin(double* out, int iters, double seed) {
    double a[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) a[i] = seed + threadIdx.x * 1e-3 + i * 1e-2;
    const double b = 1.0 + 1e-9 * threadIdx.x, c = 1e-12;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 32 / CH; r++)
#pragma unroll
            for (int i = 0; i < CH; i++) O::op(a[i], b, c);
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < CH; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <class O, int CH> static void run(double* d, int waves_per_simd, int per_op) {
    const int blocks = 256 * waves_per_simd, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_spin<O, CH>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_spin<O, CH>), dim3(blocks), dim3(256), 0, 0, d, iters, 3.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)blocks * 4 * iters * 32 * per_op;
    printf("%-46s chains=%d w/simd=%d %8.3f ms  %.3e wave-instr/s  %.2f cyc/instr/SIMD\n", O::nm(), CH, waves_per_simd, ms,
           wave_insts / (ms * 1e-3), (ms * 1e-3) * 2.4e9 * 1024 / wave_insts);
}

int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<MulF64, 1>(d, w, 1); run<MulF64, 2>(d, w, 1); run<MulF64, 4>(d, w, 1); run<MulF64, 8>(d, w, 1);
        run<AddF64, 1>(d, w, 1); run<AddF64, 4>(d, w, 1); run<AddF64, 8>(d, w, 1);
        run<FmaF64, 1>(d, w, 1); run<FmaF64, 4>(d, w, 1); run<FmaF64, 8>(d, w, 1);
        run<MulAdd, 1>(d, w, 2); run<MulAdd, 2>(d, w, 2); run<MulAdd, 4>(d, w, 2); run<MulAdd, 8>(d, w, 2);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
