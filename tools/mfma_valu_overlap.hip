// mfma_valu_overlap.hip -- do matrix-pipe and vector-ALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// (Question behind it: vis_batch_run's matcher (k_knn_mfma: FP4 block-scaled MFMA + a top-2 chain on the vector ALU) adds its full
// stand-alone time to the step although the detect kernels beside it never touch the matrix pipe.)
// One kernel, one 1024-thread workgroup per CU = 4 waves per SIMD (wave i of a workgroup runs on SIMD i % 4): the first NM of the four
// waves of every SIMD spin on v_mfma_scale_f32_32x32x64_f8f6f4 (four independent accumulators), the others on a vector chain (full-rate
// class add + xor, or half-rate class pk_min + pk_max).  Times: matrix waves alone (the others exit at once), vector waves alone, both.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o vi-slam_amd/lib/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int VKIND, int MKIND>
__global__ __launch_bounds__(1024) void k_mix(uint32_t* out, int it_m, int it_v, int nm, int run_m, int run_v, uint32_t seed) {
    const bool matrix = (int)(threadIdx.x >> 8) < nm;                 // waves 4 s .. 4 s + 3 are slot s of SIMDs 0 .. 3
    uint32_t r = 0;
    if (matrix) {
        if (!run_m) return;
        v8i a, b;
#pragma unroll
        for (int i = 0; i < 8; i++) { a[i] = (int)(seed * 0x9E3779B9u + threadIdx.x * 31 + i); b[i] = (int)(seed + threadIdx.x * 17 + i * 5); }
        v16f acc[4];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
        if (MKIND == 0) {
            for (int it = 0; it < it_m; it++) {
#pragma unroll
                for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[c], 4, 4, 0, 127, 0, 127);
            }
        } else {                                                    // v_mfma_i32_32x32x32_i8: half the MACs per instruction
            v16i ia[4];
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int i = 0; i < 16; i++) ia[c][i] = 0;
            const v4i a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
            for (int it = 0; it < it_m; it++) {
#pragma unroll
                for (int c = 0; c < 4; c++) ia[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, ia[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int i = 0; i < 16; i++) r += (uint32_t)ia[c][i];
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 16; i++) r += (uint32_t)acc[c][i];
    } else {
        if (!run_v) return;
        uint32_t x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = seed + threadIdx.x * 7 + i * 13 + blockIdx.x;
        for (int it = 0; it < it_v; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (VKIND == 0) x[i] = (x[i] + x[(i + 1) & 7]) ^ seed;
                else {
                    typedef short pk __attribute__((ext_vector_type(2)));
                    pk p = __builtin_bit_cast(pk, x[i]), q = __builtin_bit_cast(pk, x[(i + 1) & 7]);
                    p = __builtin_elementwise_min(p, q); p = __builtin_elementwise_max(p, __builtin_bit_cast(pk, x[(i + 5) & 7]));
                    x[i] = __builtin_bit_cast(uint32_t, p);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) r += x[i];
    }
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

static int g_blocks = 256;
template <int VKIND, int MKIND> static float run(uint32_t* d, int it_m, int it_v, int nm, int rm, int rv) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_mix<VKIND, MKIND>), dim3(g_blocks), dim3(1024), 0, 0, d, 10, 10, nm, rm, rv, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<VKIND, MKIND>), dim3(g_blocks), dim3(1024), 0, 0, d, it_m, it_v, nm, rm, rv, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int VKIND, int MKIND> static void table(const char* name, uint32_t* d) {
    printf("matrix: %s; vector chain: %s, %d workgroups (one per CU)\n", MKIND ? "v_mfma_i32_32x32x32_i8" : "v_mfma_scale_f32_32x32x64_f8f6f4 (FP4)", name, g_blocks);
    printf("%-28s %10s %10s %10s %10s\n", "matrix waves per SIMD (of 4)", "matrix ms", "vector ms", "both ms", "both / (m + v)");
    for (int nm : {1, 2, 3}) {
        // iteration counts chosen so that each side alone takes a comparable time
        const int it_m = 20000 / nm, it_v = 60000 / (4 - nm) * 2;
        const float m = run<VKIND, MKIND>(d, it_m, it_v, nm, 1, 0), v = run<VKIND, MKIND>(d, it_m, it_v, nm, 0, 1), b = run<VKIND, MKIND>(d, it_m, it_v, nm, 1, 1);
        printf("%-28d %10.3f %10.3f %10.3f %10.2f   (max(m, v) = %.3f)\n", nm, m, v, b, b / (m + v), m > v ? m : v);
    }
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 1024 * 4);
    // 256 workgroups = the whole chip; 16 = two CUs per XCD (a power / clock limit would not bind there)
    for (int blocks : {256, 16}) {
        g_blocks = blocks;
        table<0, 0>("v_add_u32 + v_xor_b32 (full-rate class)", d);
        table<1, 0>("v_pk_min_i16 + v_pk_max_i16 (half-rate class)", d);
        table<0, 1>("v_add_u32 + v_xor_b32 (full-rate class)", d);
    }
    hipFree(d);
    return 0;
}
