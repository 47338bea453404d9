// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands on gfx950: (1) operand packing / exactness check with
// +-1 data against a host dot product, (2) issue rate against v_mfma_i32_32x32x32_i8.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A, B: [32][32 bytes] = 64 nibbles per row (element k of a row = nibble k & 1 of byte k >> 1, low nibble first)
__global__ void k_check(const uint8_t* A, const uint8_t* B, float* D, float cinit) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const v4i a4 = *reinterpret_cast<const v4i*>(A + r * 32 + 16 * h);
    const v4i b4 = *reinterpret_cast<const v4i*>(B + r * 32 + 16 * h);
    const v8i a = {a4[0], a4[1], a4[2], a4[3], 0, 0, 0, 0}, b = {b4[0], b4[1], b4[2], b4[3], 0, 0, 0, 0};
    v16f c;
    for (int i = 0; i < 16; i++) c[i] = cinit + (float)i;
    // cbsz = blgp = 4 (fp4 e2m1); scales: e8m0 133 = 2^6 on both sides -> products +-4096
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 133, 0, 133);
    for (int i = 0; i < 16; i++) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

template <int MODE>
__global__ void k_rate(int iters, float* out) {
    v8i a = {0x22222222, 0x2A2A2A2A, (int)0xA2A2A2A2, 0x22222222, 0, 0, 0, 0}, b = a;
    v4i ai = {0x40404040, 0x40404040, 0x40404040, 0x40404040}, bi = ai;
    v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
    v16i d0 = {}, d1 = {}, d2 = {}, d3 = {};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 127, 0, 127);
        } else {
            d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, d3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i] + (float)(d0[i] + d1[i] + d2[i] + d3[i]);
    if (s == 12345.f) out[0] = s;
}

int main() {
    std::vector<uint8_t> A(32 * 32), B(32 * 32);
    std::vector<int> sa(32 * 64), sb(32 * 64);
    srand(7);
    for (int r = 0; r < 32; r++)
        for (int k = 0; k < 64; k++) {
            sa[r * 64 + k] = (rand() & 1) ? 1 : -1; sb[r * 64 + k] = (rand() & 1) ? 1 : -1;
            const uint8_t na = sa[r * 64 + k] > 0 ? 0x2 : 0xA, nb = sb[r * 64 + k] > 0 ? 0x2 : 0xA;
            if (k & 1) { A[r * 32 + k / 2] |= na << 4; B[r * 32 + k / 2] |= nb << 4; } else { A[r * 32 + k / 2] = na; B[r * 32 + k / 2] = nb; }
        }
    uint8_t *dA, *dB; float *dD, *dO;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4); hipMalloc(&dO, 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    const float cinit = 1048576.f + 4097.f;
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, dA, dB, dD, cinit);
    std::vector<float> D(32 * 32);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int dot = 0;
            for (int k = 0; k < 64; k++) dot += sa[i * 64 + k] * sb[j * 64 + k];
            // register index of (row i) inside its lane: rows (reg&3)+8*(reg>>2)+4h
            const int reg = (i & 3) + 4 * ((i >> 3));
            const float want = cinit + (float)reg + 4096.f * dot;
            if (D[i * 32 + j] != want) { if (bad < 5) printf("mismatch D[%d][%d] = %.1f want %.1f\n", i, j, D[i * 32 + j], want); bad++; }
        }
    printf("fp4 32x32x64 check: %d mismatches of 1024 (exact +-4096 products on top of a 2^20 start)\n", bad);
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000, blocks = 256 * 4 * 2;         // 2 waves per SIMD
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, 10, dO); else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, 10, dO);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, iters, dO); else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, iters, dO);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)blocks * iters * 4;
        printf("%s: %.3f ms, %.1f cycles per MFMA per SIMD at 2.4 GHz, %.2f P MAC-ops/s\n", mode == 0 ? "fp4 32x32x64 (scaled)" : "i8 32x32x32",
               ms, ms * 1e-3 * 2.4e9 / (n / 1024), n * 32 * 32 * (mode == 0 ? 64 : 32) * 2 / (ms * 1e-3) / 1e15);
    }
    return bad ? 1 : 0;
}
