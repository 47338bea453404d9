// valu_cnd.hip -- issue cost of v_cndmask_b32 variants on gfx950 (8 waves per SIMD, 8 independent chains per wave).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_cnd.hip -o vi-slam_amd/lib/valu_cnd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k_spin(uint32_t* out, int iters, uint32_t seed, unsigned long long mask) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i * 13 + blockIdx.x;
    const uint32_t b = seed * 3 + threadIdx.x;
    if (MODE == 3) asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));        // SGPR-pair selector
                if (MODE == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                        // vcc, same dst/src0
                if (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "s"(mask));        // src order swapped
                if (MODE == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                        // vcc initialised
                if (MODE == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                 // reference: full-rate op
                if (MODE == 5) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                              // reference: half-rate op
                if (MODE == 6) { a[i] = ((mask >> (i + r)) & 1) ? a[i] ^ b : a[i] + b; }                                    // compiler-generated uniform select
                if (MODE == 7) asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(a[i]) : "v"(b), "v"(seed));                  // bitfield select as a cndmask replacement
                if (MODE == 8) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 9) asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(a[i]) : "s"((uint32_t)mask));
                if (MODE == 10) asm volatile("v_cmp_gt_i16_sdwa s[20:21], %0, %1 src0_sel:WORD_1 src1_sel:DWORD" :: "v"(a[i]), "v"(b) : "s20", "s21");
                if (MODE == 11) asm volatile("v_cmp_lt_i16_e64 s[20:21], %0, %1" :: "v"(a[i]), "v"(b) : "s20", "s21");
                if (MODE == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                  // e64 encoding, vcc as the explicit selector
                if (MODE == 14) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");      // carry-in from vcc (e32)
                if (MODE == 15) asm volatile("v_addc_co_u32_e64 %0, s[20:21], %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask) : "s20", "s21");
                if (MODE == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_pk_min_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));   // 1 + 3 mix
                if (MODE == 17) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2\n v_pk_min_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1\n v_pk_sub_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b), "s"(mask));
                if (MODE == 18) asm volatile("v_cndmask_b32_sdwa %0, %0, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(a[i]) : "v"(b));
                if (MODE == 19) asm volatile("v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (MODE == 20) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc\n v_pk_min_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 21) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %0, %1, %0, vcc\n v_pk_min_i16 %0, %0, %1\n v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 22) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b));
                if (MODE == 23) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (MODE == 25) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(*(unsigned long long*)&a[i & 6]) : "v"(b), "v"(seed) : "s20", "s21");
                if (MODE == 30) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(1.0000001));
                if (MODE == 31) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&a[i & 6]) : "v"(1.0000001));
                if (MODE == 32) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(*(double*)&a[i & 6]) : "v"(1.0000001));
                if (MODE == 33) asm volatile("v_rndne_f64 %0, %0" : "+v"(*(double*)&a[i & 6]));
                if (MODE == 34) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*(double*)&a[i & 6]) : "v"(b));
                if (MODE == 35) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(1.0000001));
                if (MODE == 36) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 37) asm volatile("v_rcp_f64 %0, %0" : "+v"(*(double*)&a[i & 6]));
                if (MODE == 38) asm volatile("v_sqrt_f64 %0, %0" : "+v"(*(double*)&a[i & 6]));
                if (MODE == 39) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(*(double*)&a[i & 6]) : "v"(1.0000001) : "vcc");
                if (MODE == 12) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
            }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE> static void run(uint32_t* d, const char* name, unsigned long long mask) {
    const int blocks = 256 * 8, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_spin<MODE>, dim3(blocks), dim3(256), 0, 0, d, 100, 1u, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_spin<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)blocks * 4 * iters * 32;
    printf("%-52s %8.3f ms  %.2f cyc/instr/SIMD\n", name, ms, (ms * 1e-3) * 2.4e9 * 1024 / wave_insts);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (unsigned long long m : {0xAAAA5555F0F01234ull}) {
        printf("mask %016llx\n", m);
        run<0>(d, "v_cndmask_b32_e64 a, a, b, s[..]", m);
        run<2>(d, "v_cndmask_b32_e64 a, b, a, s[..]", m);
        run<1>(d, "v_cndmask_b32 a, a, b, vcc (vcc as found)", m);
        run<3>(d, "v_cndmask_b32 a, a, b, vcc (vcc = mask)", m);
        run<6>(d, "compiler select on a uniform bit", m);
        run<7>(d, "v_bfi_b32", m);
        run<9>(d, "v_mbcnt_hi_u32_b32 a, s, a", m);
        run<10>(d, "v_cmp_gt_i16_sdwa s[20:21]", m);
        run<11>(d, "v_cmp_lt_i16_e64 s[20:21]", m);
        run<12>(d, "v_cmp_lt_i32_e32 vcc", m);
        run<13>(d, "v_cndmask_b32_e64 a, a, b, vcc", m);
        run<14>(d, "v_addc_co_u32_e32 (vcc in/out)", m);
        run<15>(d, "v_addc_co_u32_e64 (sgpr pair in/out)", m);
        run<16>(d, "[cndmask vcc + 3 pk ops] / 4", m);
        run<17>(d, "[cndmask_e64 sgpr + 3 pk ops] / 4", m);
        run<20>(d, "[2 cndmask vcc + 2 pk ops] per block of 4", m);
        run<21>(d, "[2 cndmask_e64 vcc + 2 pk ops] per block of 4", m);
        run<22>(d, "[4 cndmask vcc, dependent] per block of 4", m);
        run<23>(d, "v_mul_hi_u32", m);
        run<24>(d, "v_mul_hi_u32_u24", m);
        run<30>(d, "v_mul_f64 (4 chains)", m);
        run<31>(d, "v_add_f64 (4 chains)", m);
        run<32>(d, "v_fma_f64 (4 chains)", m);
        run<33>(d, "v_rndne_f64", m);
        run<34>(d, "v_cvt_f64_f32", m);
        run<35>(d, "v_cvt_f32_f64", m);
        run<36>(d, "v_rcp_f32", m);
        run<37>(d, "v_rcp_f64", m);
        run<38>(d, "v_sqrt_f64", m);
        run<39>(d, "v_div_scale_f64", m);
        run<18>(d, "v_cndmask_b32_sdwa vcc", m);
        run<19>(d, "v_cndmask_b32_dpp vcc", m);
        run<4>(d, "v_xor_b32 (reference)", m);
        run<8>(d, "v_and_b32 (reference)", m);
        run<5>(d, "v_pk_min_i16 (reference)", m);
    }
    hipFree(d);
    return 0;
}
