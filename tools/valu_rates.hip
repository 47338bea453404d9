// valu_rates.hip -- issue rate of individual VALU instructions on gfx950, 8 waves per SIMD, 8 independent
// dependency chains per wave.  Prints cycles per wave64 instruction per SIMD (2.4 GHz nominal).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o vi-slam_amd/lib/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define OP3(name, text)                                                                                   \
    struct name { static __device__ __forceinline__ void op(uint32_t& a, uint32_t b, uint32_t c) {      \
        asm volatile(text : "+v"(a) : "v"(b), "v"(c)); } static const char* nm() { return text; } };

OP3(AddU32,     "v_add_u32 %0, %0, %1")
OP3(AndB32,     "v_and_b32 %0, %0, %1")
OP3(MinI32,     "v_min_i32 %0, %0, %1")
OP3(Min3I32,    "v_min3_i32 %0, %0, %1, %2")
OP3(Add3U32,    "v_add3_u32 %0, %0, %1, %2")
OP3(LshlOr,     "v_lshl_or_b32 %0, %0, 3, %1")
OP3(PermB32,    "v_perm_b32 %0, %0, %1, %2")
OP3(AlignByte,  "v_alignbyte_b32 %0, %0, %1, 1")
OP3(PkMinI16,   "v_pk_min_i16 %0, %0, %1")
OP3(PkSubI16,   "v_pk_sub_i16 %0, %0, %1")
OP3(PkMinF16,   "v_pk_min_f16 %0, %0, %1")
OP3(PkAddF16,   "v_pk_add_f16 %0, %0, %1")
OP3(PkMax3F16,  "v_pk_maximum3_f16 %0, %0, %1, %2")
OP3(MinF32,     "v_min_f32 %0, %0, %1")
OP3(AddF32,     "v_add_f32 %0, %0, %1")
OP3(FmaF32,     "v_fma_f32 %0, %0, %1, %2")
OP3(Max3F32,    "v_max3_f32 %0, %0, %1, %2")
OP3(Min3F16,    "v_min3_f16 %0, %0, %1, %2")
OP3(Dot4U8,     "v_dot4_u32_u8 %0, %1, %2, %0")
OP3(Dot2U16,    "v_dot2_u32_u16 %0, %1, %2, %0")
OP3(MulU24,     "v_mul_u32_u24 %0, %0, %1")
OP3(MadU24,     "v_mad_u32_u24 %0, %0, %1, %2")
OP3(MulLo,      "v_mul_lo_u32 %0, %0, %1")
OP3(SadU8,      "v_sad_u8 %0, %1, %2, %0")
OP3(Bcnt,       "v_bcnt_u32_b32 %0, %1, %0")
OP3(CvtF32U8,   "v_cvt_f32_ubyte0 %0, %1")
OP3(Bfe,        "v_bfe_u32 %0, %0, 3, 8")
OP3(MinU16,     "v_min_u16 %0, %0, %1")
OP3(XorB32,     "v_xor_b32 %0, %0, %1")
OP3(SubU32,     "v_sub_u32 %0, %0, %1")
OP3(Lshl,       "v_lshlrev_b32 %0, 3, %0")
OP3(Lshr,       "v_lshrrev_b32 %0, 1, %0")
OP3(MaxU32,     "v_max_u32 %0, %0, %1")
OP3(MinI16,     "v_min_i16 %0, %0, %1")
OP3(MaxI16,     "v_max_i16 %0, %0, %1")
OP3(AddU16,     "v_add_u16 %0, %0, %1")
OP3(SubU16,     "v_sub_u16 %0, %0, %1")
OP3(MinU16Sdwa, "v_min_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")
OP3(SubU16Sdwa, "v_sub_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3")
OP3(AddU32Sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_3")
OP3(MulF32,     "v_mul_f32 %0, %0, %1")
OP3(MaxF32,     "v_max_f32 %0, %0, %1")
OP3(PkAddU16,   "v_pk_add_u16 %0, %0, %1")
OP3(PkMaxI16,   "v_pk_max_i16 %0, %0, %1")
OP3(PkMulLoU16, "v_pk_mul_lo_u16 %0, %0, %1")
OP3(Cndmask,    "v_cndmask_b32 %0, %0, %1, vcc")
OP3(MovB32,     "v_mov_b32 %0, %1")
OP3(Bfi,        "v_bfi_b32 %0, %0, %1, %2")
OP3(AndOr,      "v_and_or_b32 %0, %0, %1, %2")
OP3(LshlAdd,    "v_lshl_add_u32 %0, %0, 2, %1")
OP3(MadI24,     "v_mad_i32_i24 %0, %0, %1, %2")
OP3(CvtI32F32,  "v_cvt_i32_f32 %0, %0")
OP3(RndneF32,   "v_rndne_f32 %0, %0")
OP3(Mbcnt,      "v_mbcnt_lo_u32_b32 %0, %1, %0")
OP3(AddF16,     "v_add_f16 %0, %0, %1")
OP3(MaxF16,     "v_max_f16 %0, %0, %1")
OP3(CmpLtI32,   "v_cmp_lt_i32 vcc, %0, %1")
OP3(MulF64x,    "v_mul_f64 %0, %0, %0")

template <class O>
__global__ __launch_bounds__(256) void k_spin(uint32_t* out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 7 + i * 13 + blockIdx.x;
    const uint32_t b = seed * 3 + threadIdx.x, c = 0x0c020c01u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) O::op(a[i], b, c);
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <class O> static void run(uint32_t* d, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_spin<O>, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_spin<O>, dim3(blocks), dim3(256), 0, 0, d, iters, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)blocks * 4 * iters * 32;
    printf("%-36s w/simd=%d %8.3f ms  %.3e wave-instr/s  %.2f cyc/instr/SIMD\n", O::nm(), waves_per_simd, ms,
           wave_insts / (ms * 1e-3), (ms * 1e-3) * 2.4e9 * 1024 / wave_insts);
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {2, 8}) {
        run<AddU32>(d, w); run<AndB32>(d, w); run<MinI32>(d, w); run<Min3I32>(d, w); run<Add3U32>(d, w); run<LshlOr>(d, w);
        run<PermB32>(d, w); run<AlignByte>(d, w); run<PkMinI16>(d, w); run<PkSubI16>(d, w); run<PkMinF16>(d, w);
        run<PkAddF16>(d, w); run<PkMax3F16>(d, w); run<MinF32>(d, w); run<AddF32>(d, w); run<FmaF32>(d, w); run<Max3F32>(d, w);
        run<Min3F16>(d, w); run<Dot4U8>(d, w); run<Dot2U16>(d, w); run<MulU24>(d, w); run<MadU24>(d, w); run<MulLo>(d, w);
        run<SadU8>(d, w); run<Bcnt>(d, w); run<CvtF32U8>(d, w); run<Bfe>(d, w); run<MinU16>(d, w);
        run<XorB32>(d, w); run<SubU32>(d, w); run<Lshl>(d, w); run<Lshr>(d, w); run<MaxU32>(d, w); run<MinI16>(d, w); run<MaxI16>(d, w);
        run<AddU16>(d, w); run<SubU16>(d, w); run<MinU16Sdwa>(d, w); run<SubU16Sdwa>(d, w); run<AddU32Sdwa>(d, w); run<MulF32>(d, w); run<MaxF32>(d, w);
        run<PkAddU16>(d, w); run<PkMaxI16>(d, w); run<PkMulLoU16>(d, w); run<Cndmask>(d, w); run<MovB32>(d, w); run<Bfi>(d, w); run<AndOr>(d, w);
        run<LshlAdd>(d, w); run<MadI24>(d, w); run<CvtI32F32>(d, w); run<RndneF32>(d, w); run<Mbcnt>(d, w); run<AddF16>(d, w); run<MaxF16>(d, w);
        run<CmpLtI32>(d, w);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
