// valu_peak.hip -- measures the achievable VALU issue rate on this GPU for the instruction kinds the
// detect/match kernels are made of (32-bit integer, packed 16-bit, f32 fma), so roofline claims for the
// compute-bound kernels (k_fast, k_knn2, k_describe) can be priced against a MEASURED ceiling.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o vi-slam_amd/lib/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void k_spin(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i * 13 + blockIdx.x;
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = (float)(a[i] & 255) * 0.001f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) { a[i] = __popc(a[i] ^ seed) + a[(i + 1) & 7]; }                 // v_xor + v_bcnt(acc)
            else if (KIND == 1) { a[i] = min(a[i] ^ 0x5bd1e995u, a[(i + 3) & 7]) + 1u; }    // v_xor, v_min, v_add
            else if (KIND == 2) {                                                            // v_pk_min_i16 / v_pk_max_i16
                typedef short pk __attribute__((ext_vector_type(2)));
                pk x = __builtin_bit_cast(pk, a[i]), y = __builtin_bit_cast(pk, a[(i + 1) & 7]);
                x = __builtin_elementwise_min(x, y); x = __builtin_elementwise_max(x, __builtin_bit_cast(pk, seed));
                a[i] = __builtin_bit_cast(uint32_t, x) + 0x00010001u;
            } else if (KIND == 4) { a[i] = (a[i] + a[(i + 1) & 7]) ^ seed; }                // v_add_u32 + v_xor_b32: the full-rate class
            else if (KIND == 5) {                                                            // v_pk_min_i16 + v_pk_max_i16: the half-rate class
                typedef short pk __attribute__((ext_vector_type(2)));
                pk x = __builtin_bit_cast(pk, a[i]), y = __builtin_bit_cast(pk, a[(i + 1) & 7]);
                x = __builtin_elementwise_min(x, y); x = __builtin_elementwise_max(x, __builtin_bit_cast(pk, a[(i + 5) & 7]));
                a[i] = __builtin_bit_cast(uint32_t, x);
            } else { f[i] = fmaf(f[i], 1.0001f, f[(i + 1) & 7]); }                         // v_fma_f32
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + (uint32_t)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

static double g_rate[8];
template <int KIND> static void run(const char* name, int ops_per_inner, uint32_t* d) {
    const int blocks = 256 * 8, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_spin<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)blocks * 4 * iters * 8 * ops_per_inner;
    g_rate[KIND] = wave_insts / (ms * 1e-3);
    printf("%-28s %8.3f ms  %.3e wave-instr/s  = %.1f T lane-ops/s  (%.2f cycles/instr/SIMD at 2.4 GHz, 1024 SIMDs)\n", name, ms,
           wave_insts / (ms * 1e-3), wave_insts * 64 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 * 1024 / wave_insts);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("xor+bcnt (2 instr)", 2, d);
    run<1>("xor+min+add (3 instr)", 3, d);
    run<2>("pk_min+pk_max+add (3 instr)", 3, d);
    run<3>("fma_f32 (1 instr)", 1, d);
    run<4>("add+xor (2 instr, full rate)", 2, d);
    run<5>("pk_min+pk_max (2 instr, half rate)", 2, d);
    hipFree(d);
    // one JSON line for tools/pmc_summarize.py -> profiles/pmc_traffic.json["valu_peak_measured"] (bench.py prices the detect kernels against it)
    printf("{\"valu_peak_measured\": {\"unit\": \"wave-instr/s\", \"full_rate_class\": %.4e, \"half_rate_class\": %.4e, \"mixed_pk_min_max_add\": %.4e, "
           "\"what\": \"tools/valu_peak.hip on the profiling box: 8 waves per SIMD, independent chains; full = v_add_u32 + v_xor_b32, half = v_pk_min_i16 + "
           "v_pk_max_i16, mixed = 2 half + 1 full\"}}\n", g_rate[4], g_rate[5], g_rate[2]);
    return 0;
}
