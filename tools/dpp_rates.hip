// dpp_rates.hip -- issue rate of the DPP wave shifts and the other instructions the streaming k_fast is made of (gfx950), at 2 / 4 / 8
// waves per SIMD, 8 independent chains per wave; also checks what wave_shr:1 / wave_shl:1 deliver in lanes 0 / 63 and across the rows.
// build: hipcc --offload-arch=gfx950 -O3 tools/dpp_rates.hip -o vi-slam_amd/lib/dpp_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define OP3(name, text)                                                                                   \
    struct name { static __device__ __forceinline__ void op(uint32_t& a, uint32_t b, uint32_t c) {      \
        asm volatile(text : "+v"(a) : "v"(b), "v"(c)); } static const char* nm() { return text; } };
OP3(AddU32,     "v_add_u32 %0, %0, %1")
OP3(MovB32,     "v_mov_b32 %0, %1")
OP3(MovWaveShr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
OP3(MovWaveShl, "v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
OP3(MovRowShr,  "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
OP3(AddRowShr,  "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
OP3(SubWaveShr, "v_sub_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
OP3(AlignByte,  "v_alignbyte_b32 %0, %0, %1, 1")
OP3(Bitop3,     "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
OP3(AndOr,      "v_and_or_b32 %0, %0, %1, %2")
OP3(Or3,        "v_or3_b32 %0, %0, %1, %2")
OP3(Ffbl,       "v_ffbl_b32 %0, %0")
OP3(Bcnt,       "v_bcnt_u32_b32 %0, %1, %0")
OP3(CndVcc,     "v_cndmask_b32 %0, %0, %1, vcc")
OP3(PkMinI16,   "v_pk_min_i16 %0, %0, %1")
OP3(PkMadI16,   "v_pk_mad_i16 %0, %0, %1, %2")
OP3(LshlOr16,   "v_lshl_or_b32 %0, %0, 16, %1")
OP3(PermB32,    "v_perm_b32 %0, %0, %1, %2")
OP3(MinU32,     "v_min_u32 %0, %0, %1")
OP3(Lshr,       "v_lshrrev_b32 %0, 1, %0")
OP3(SubU32,     "v_sub_u32 %0, %0, %1")
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
    printf("%-84s w/simd=%d %8.3f ms  %.3e wave-instr/s  %.2f cyc/instr/SIMD\n", O::nm(), waves_per_simd, ms,
           wave_insts / (ms * 1e-3), (ms * 1e-3) * 2.4e9 * 1024 / wave_insts);
}
__global__ void k_sem(int* o) {
    const int v = 100 + (int)threadIdx.x;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);          // wave_shr:1
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true);     // wave_shl:1
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    int* s; hipMalloc(&s, 128 * 4); int h[128];
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, s); hipMemcpy(h, s, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++) { if (h[i] != (i ? 99 + i : 0)) bad++; if (h[64 + i] != (i < 63 ? 101 + i : 0)) bad++; }
    printf("wave_shr:1 lane i <- lane i-1 (0 into lane 0), wave_shl:1 lane i <- lane i+1 (0 into lane 63): %s (lanes 0,1,16,32,63: shr %d %d %d %d %d shl %d %d %d %d %d)\n",
           bad ? "MISMATCH" : "ok", h[0], h[1], h[16], h[32], h[63], h[64], h[65], h[80], h[96], h[127]);
    for (int w : {2, 4, 8}) {
        run<AddU32>(d, w); run<MovB32>(d, w); run<MovWaveShr>(d, w); run<MovWaveShl>(d, w); run<MovRowShr>(d, w); run<AddRowShr>(d, w); run<SubWaveShr>(d, w);
        run<AlignByte>(d, w); run<Bitop3>(d, w); run<AndOr>(d, w); run<Or3>(d, w); run<Ffbl>(d, w); run<Bcnt>(d, w); run<CndVcc>(d, w);
        run<PkMinI16>(d, w); run<PkMadI16>(d, w); run<LshlOr16>(d, w); run<PermB32>(d, w); run<MinU32>(d, w); run<Lshr>(d, w); run<SubU32>(d, w);
        printf("\n");
    }
    return 0;
}
