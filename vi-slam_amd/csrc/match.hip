// match.hip -- brute-force Hamming 2-NN (both directions) and the reference's match filters on gfx950.
//
// k_expand + k_knn_mfma (default) / k_knn2 (popcount fallback) replace descriptorsGPU[0/1].upload +
//          2 x cuda::DescriptorMatcher::knnMatch(k=2) (/root/reference/src/MatcherGPU.cpp:49-56; CPU twin
//          Matcher::computeMatches, /root/reference/src/Matcher.cpp:83-94).  Descriptors never leave HBM between
//          detect and match.
// k_filter fuses Matcher::computeBestMatches (/root/reference/src/Matcher.cpp:353-367):
//          nnFilter :148-169, computeSymMatches :96-144, sortMatches :329-352,
//          bestMatchesFilter :171-244 and getGoodMatches :295-303 -- O(N) instead of the
//          reference's O(N1*N2) iterator scan.
// Bit-exact against oracle/match.cpp (tests/test_match_gpu.py).
#include "vis_internal.h"
#include <cstdlib>

// One lane = one query row; the train descriptor of the current iteration is wave-uniform, so the
// compiler keeps it in SGPRs (s_load_dwordx8) and the inner loop is 8 x (v_xor, v_bcnt accumulate)
// + 3 ops of top-2 maintenance on the packed key (dist << 16 | trainIdx).  Keeping the two smallest
// keys reproduces cv::batchDistance's order: ascending distance, ties -> lower train index first.
__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ desc, const int32_t* __restrict__ nkp, int kcap,
                                              const int32_t* __restrict__ pair_q, const int32_t* __restrict__ pair_t,
                                              uint32_t* __restrict__ knn12, uint32_t* __restrict__ knn21) {
    const int pair = blockIdx.y, dir = blockIdx.z;
    const int rq = dir == 0 ? pair_q[pair] : pair_t[pair];
    const int rt = dir == 0 ? pair_t[pair] : pair_q[pair];
    if (rq < 0 || rt < 0) return;
    const int nq = min(nkp[rq], kcap), nt = min(nkp[rt], kcap);
    // two query rows per lane (q and q + blockDim): every scalar-loaded train row feeds twice the
    // VALU work, halving the scalar-load stalls per popcount and doubling the independent chains
    const int qbase = blockIdx.x * blockDim.x * 2;
    if (qbase >= nq) return;
    const int q0i = qbase + threadIdx.x, q1i = q0i + blockDim.x;
    const uint4* Q0 = reinterpret_cast<const uint4*>(desc + ((size_t)rq * kcap + min(q0i, nq - 1)) * 32);
    const uint4* Q1 = reinterpret_cast<const uint4*>(desc + ((size_t)rq * kcap + min(q1i, nq - 1)) * 32);
    const uint4 qa = Q0[0], qb = Q0[1], ra = Q1[0], rb = Q1[1];
    const uint4* T = reinterpret_cast<const uint4*>(desc + (size_t)rt * kcap * 32);
    uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu, j0 = 0xFFFFFFFFu, j1 = 0xFFFFFFFFu;
#define KNN_DIST(xa, xb, ta, tb, d_) do {                                                          \
        d_ = __popc((xa).x ^ (ta).x);                                                              \
        d_ += __popc((xa).y ^ (ta).y); d_ += __popc((xa).z ^ (ta).z); d_ += __popc((xa).w ^ (ta).w); \
        d_ += __popc((xb).x ^ (tb).x); d_ += __popc((xb).y ^ (tb).y); d_ += __popc((xb).z ^ (tb).z); \
        d_ += __popc((xb).w ^ (tb).w); } while (0)
#define KNN_STEP(ta, tb, tt) do {                                                                   \
        uint32_t da_, db_;                                                                          \
        KNN_DIST(qa, qb, ta, tb, da_); KNN_DIST(ra, rb, ta, tb, db_);                               \
        const uint32_t ka_ = (da_ << 16) | (uint32_t)(tt), kb_ = (db_ << 16) | (uint32_t)(tt);      \
        k1 = min(k1, max(k0, ka_)); k0 = min(k0, ka_);                                              \
        j1 = min(j1, max(j0, kb_)); j0 = min(j0, kb_); } while (0)
    // the train descriptors are wave-uniform (SGPRs): groups of KNN_U rows are fetched with back-to-back
    // scalar loads (s_load_dwordx16) so one load latency is paid per group, not per row; the other waves
    // of the SIMD cover it with their xor/popcount work
    constexpr int KNN_U = 8;
    const int ntU = nt & ~(KNN_U - 1);
    for (int t = 0; t < ntU; t += KNN_U) {
        uint4 cur[2 * KNN_U];
#pragma unroll
        for (int j = 0; j < 2 * KNN_U; j++) cur[j] = T[2 * t + j];
#pragma unroll
        for (int u = 0; u < KNN_U; u++) KNN_STEP(cur[2 * u], cur[2 * u + 1], t + u);
    }
    for (int t = ntU; t < nt; t++) {
        const uint4 ta = T[2 * t], tb = T[2 * t + 1];
        KNN_STEP(ta, tb, t);
    }
#undef KNN_STEP
#undef KNN_DIST
    uint32_t* outp = (dir == 0 ? knn12 : knn21) + (size_t)pair * kcap * 2;
    if (q0i < nq) { outp[2 * q0i] = k0; outp[2 * q0i + 1] = k1; }
    if (q1i < nq) { outp[2 * q1i] = j0; outp[2 * q1i + 1] = j1; }
}

// ------------------------------------------------------------------------------------------------
// MFMA formulation of the same 2-NN (default path).  Every hot kernel of this library is bound by the integer VALU issue rate
// (~0.62 T wave-instructions/s measured, tools/valu_peak.hip); the matrix pipe is idle.  Descriptor bits become FP4 (e2m1) +1 / -1,
// two to a byte (128 B per descriptor), and the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 multiplies them with both scales
// at 2^6: sum_k a_k b_k = 4096 * (256 - 2*Hamming), exact in the f32 accumulator (every partial sum is an integer below 2^22).
// gfx950 issues this instruction in the cycles of v_mfma_i32_32x32x32_i8 at twice the K (tools/mfma_f4_probe.hip: 37.6 against
// 36.2 cycles, 8.6 against 4.5 P MAC/s), so a 32 x 32 block of exact distances is a chain of 4 instead of 8 MFMAs, on half the
// operand bytes.  The block is already in key form, because the chain starts from 2^20 + (tile, register) index -- 8192 * Hamming
// lands on top of the 13 free low bits -- and the VALU only maintains the top-2 keys: 2 instructions (v_med3_i32, v_min_i32 on
// the bit patterns: non-negative floats order like integers) per distance instead of 19.
// Columns (lane & 31) = the 32 descriptors whose neighbours this wave tracks (B operand, in registers); rows = the swept set,
// staged through LDS 32 descriptors at a time (row stride 144 B: conflict-free ds_read_b128).  Results are bit-identical to
// k_knn2 (same key order): tests/test_match_gpu.py.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// 32 descriptor bits -> 32 FP4 nibbles (16 bytes): bit k of the descriptor = element k, 1 -> +1 (0x2), 0 -> -1 (0xA)
__global__ __launch_bounds__(256) void k_expand(const uint8_t* __restrict__ desc, const int32_t* __restrict__ nkp, int kcap,
                                                int8_t* __restrict__ X, int rec_first, int rec_count) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long dsc = gid >> 3;
    const int part = (int)(gid & 7);
    const int rec_i = (int)(dsc / kcap), i = (int)(dsc - (long long)rec_i * kcap);
    if (rec_i >= rec_count) return;
    const int rec = rec_first + rec_i;
    if (i >= min(nkp[rec], kcap)) return;
    const uint32_t bits = *reinterpret_cast<const uint32_t*>(desc + ((size_t)rec * kcap + i) * 32 + 4 * part);
    uint32_t o[4];
#pragma unroll
    for (int n = 0; n < 4; n++) {
        uint32_t x = (bits >> (8 * n)) & 0xFFu;                   // 8 bits -> bit 4 k of a dword
        x = (x | (x << 12)) & 0x000F000Fu;
        x = (x | (x << 6)) & 0x03030303u;
        x = (x | (x << 3)) & 0x11111111u;
        o[n] = 0x22222222u | ((~x & 0x11111111u) << 3);          // magnitude 1, sign bit set where the descriptor bit is 0
    }
    *reinterpret_cast<uint4*>(X + ((size_t)rec * kcap + i) * 128 + 16 * part) = make_uint4(o[0], o[1], o[2], o[3]);
}

#define KM_ROW 9             // uint4 per LDS row: 128 B of descriptor + 16 B pad
__device__ __forceinline__ int imed3(int a, int b, int c) {
    int d;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// NC = column groups of 32 fixed descriptors per wave.  NC = 2: every 16-byte LDS read of a swept row feeds two MFMA chains
// (the LDS pipe and the staging traffic per MFMA halve, one barrier per 8 MFMAs instead of 4) at the price of 32 + 32
// operand / accumulator registers per lane.
template <int NC>
__global__ __launch_bounds__(256) void k_knn_mfma(const int8_t* __restrict__ X, const int32_t* __restrict__ nkp, int kcap,
                                                  const int32_t* __restrict__ pair_q, const int32_t* __restrict__ pair_t,
                                                  uint32_t* __restrict__ knn12, uint32_t* __restrict__ knn21, int npairs, int nchunks) {
    __shared__ __attribute__((aligned(16))) uint4 tile[2][32 * KM_ROW];
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (private L2 each).  All chunks and both
    // directions of one frame pair go to ONE XCD, so the pair's two descriptor sets are fetched into one L2 once
    // (rocprofv3 FETCH_SIZE was 2.3 GB per 512 pairs with a plain (chunk, pair, dir) grid: every chunk re-fetched the
    // swept set through a different L2).
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, per_pair = 2 * nchunks;
    const int pair = (jb / per_pair) * 8 + xcd;
    if (pair >= npairs) return;
    const int inner = jb % per_pair, dir = inner / nchunks, chunk = inner - dir * nchunks;
    const int rf = dir == 0 ? pair_q[pair] : pair_t[pair];        // fixed set: top-2 tracked per descriptor
    const int rs = dir == 0 ? pair_t[pair] : pair_q[pair];        // swept set
    if (rf < 0 || rs < 0) return;
    const int nf = min(nkp[rf], kcap), ns = min(nkp[rs], kcap);
    const int fbase = chunk * (128 * NC);
    if (fbase >= nf) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int fidx0 = fbase + wave * (32 * NC) + col;
    // operand layout of the 32x32x64 FP4 form (probed with exact data): lane (r, h) holds elements k = 32 h .. 32 h + 31 of row /
    // column r as 16 bytes, low nibble first; K step ks covers descriptor bits 64 ks .. 64 ks + 63
    v8i bf[NC][4];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const int8_t* xf = X + ((size_t)rf * kcap + min(fidx0 + 32 * c, nf - 1)) * 128 + 16 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            // the fixed operand is negated (x ^ 0x8 per nibble flips the FP4 sign), so the chain adds -dot =
            // 4096 * (2 * Hamming - 256) onto its start value
            const v4i x = *reinterpret_cast<const v4i*>(xf + 32 * ks);
            bf[c][ks] = v8i{x[0] ^ (int)0x88888888, x[1] ^ (int)0x88888888, x[2] ^ (int)0x88888888, x[3] ^ (int)0x88888888, 0, 0, 0, 0};
        }
    }
    const int8_t* xs = X + (size_t)rs * kcap * 128;
    const int ntiles = (ns + 31) / 32;
    // staging of the next swept tile (32 rows x 128 B = one uint4 per thread): the global load is issued before the MFMA chain of
    // the current tile, the LDS write after it (the load latency is covered by the wave's own compute, not only by other waves)
    const int st_row = tid >> 3, st_c = tid & 7;
    uint4 pre;
#define STAGE_LOAD(t_) do { pre = *reinterpret_cast<const uint4*>(xs + (size_t)min((t_) * 32 + st_row, ns - 1) * 128 + 16 * st_c); } while (0)
#define STAGE_STORE(buf_) do { tile[buf_][st_row * KM_ROW + st_c] = pre; } while (0)
    // The dot products are multiples of 8192 after the 2^20 offset, so the 13 low bits of an accumulator are free: every chain STARTS
    // from 2^20 + 8192 + accumulator register, and the finished accumulator IS the key 8192 * (Hamming + 1) + register of a row of the
    // CURRENT tile -- no per-element key construction.  The tile index is carried by AGE instead of by the start values (round 6: 16
    // v_add_f32 per tile on the start values before, 2 v_sub_f32 per column group now): in front of every tile's merge the two running
    // keys lose 16, so a key that is D tiles old reads 8192 * (Hamming + 1) + register - 16 D.  Inside a lane the register order is the
    // row order and an older tile means lower rows, so among equal distances the smaller key is still the lower row; 16 D <= 8176 never
    // borrows across a distance (ns <= 16384: D <= 511) and never goes negative.  At the end w = key + 16 (ntiles - 1) restores
    // 8192 * (Hamming + 1) + (tile << 4 | register).  Keys are compared as the bit patterns of positive floats; KNN_NONE (1.7e38: a
    // finite float that 16 cannot change) = no neighbour yet.
#define KNN_NONE 0x7F000000
    int k0[NC], k1[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { k0[c] = KNN_NONE; k1[c] = KNN_NONE; }
    v16f cstart;
#pragma unroll
    for (int r = 0; r < 16; r++) cstart[r] = 1048576.f + 8192.f + (float)r;
    if (ntiles > 0) { STAGE_LOAD(0); STAGE_STORE(0); }
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        if (t + 1 < ntiles) STAGE_LOAD(t + 1);
        const uint4* tb = tile[t & 1] + col * KM_ROW + h;
        v16f acc[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = cstart;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const uint4 au = tb[2 * ks];
            const v8i a = {(int)au.x, (int)au.y, (int)au.z, (int)au.w, 0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NC; c++)      // cbsz = blgp = 4: FP4 e2m1 on both sides; scales 2^6 (e8m0 133): products +-4096
                acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bf[c][ks], acc[c], 4, 4, 0, 133, 0, 133);
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {                                 // age the running keys by one tile (see above); exact float subtractions
            k0[c] = __float_as_int(__int_as_float(k0[c]) - 16.f);
            k1[c] = __float_as_int(__int_as_float(k1[c]) - 16.f);
        }
        if (t * 32 + 32 <= ns) {
            // two keys per step, three instructions (1.5 per key instead of 2): the second smallest of {k0, k1, x, y} is k1 or the
            // second smallest of {k0, x, y} (k1 >= k0 >= the smallest of those three); all keys are distinct (they carry their tile
            // and register), so the order in which they are merged cannot change the result
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int x = __float_as_int(acc[c][r]), y = __float_as_int(acc[c][r + 1]);
                    k1[c] = min(k1[c], imed3(k0[c], x, y));
                    k0[c] = min(min(k0[c], x), y);
                }
        } else {
            const int toff = t * 32 + 4 * h;
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int trow = toff + (r & 3) + 8 * (r >> 2);
                    const int key = trow < ns ? __float_as_int(acc[c][r]) : KNN_NONE;
                    k1[c] = imed3(k0[c], k1[c], key);
                    k0[c] = min(k0[c], key);
                }
        }
        if (t + 1 < ntiles) STAGE_STORE((t + 1) & 1);
        __syncthreads();
    }
#undef STAGE_LOAD
#undef STAGE_STORE
    // (Hamming << 16) | row: the record format of k_filter / the popcount kernel
    auto true_key = [&](int k) -> uint32_t {
        const uint32_t w = (uint32_t)__int_as_float(k) + 16u * (uint32_t)(ntiles - 1);      // the float IS the integer key (exact); undo the ageing
        const uint32_t lo = w & 8191u, r = lo & 15u;
        const uint32_t row = (lo >> 4) * 32u + 4u * (uint32_t)h + (r & 3u) + 8u * (r >> 2);
        return k == KNN_NONE ? 0xFFFFFFFFu : ((((w >> 13) - 1u) << 16) | row);
    };
#undef KNN_NONE
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const uint32_t a0 = true_key(k0[c]), a1 = true_key(k1[c]);
        // lanes l and l+32 hold the two row halves of the same column
        const uint32_t o0 = __shfl_xor(a0, 32), o1 = __shfl_xor(a1, 32);
        const uint32_t m0 = min(a0, o0), m1 = min(max(a0, o0), min(a1, o1));
        const int fidx = fidx0 + 32 * c;
        if (h == 0 && fidx < nf) {
            uint32_t* outp = (dir == 0 ? knn12 : knn21) + ((size_t)pair * kcap + fidx) * 2;
            outp[0] = m0; outp[1] = m1;
        }
    }
}

__device__ __forceinline__ uint32_t fmap_f(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ bool ratio_survives(uint32_t k0, uint32_t k1, double ratio) {
    if (k0 == 0xFFFFFFFFu || k1 == 0xFFFFFFFFu) return false;               // fewer than 2 neighbours
    const double d0 = (double)(float)(k0 >> 16), d1 = (double)(float)(k1 >> 16);
    return !(d0 > ratio * d1);                                              // src/Matcher.cpp:158
}

// block per pair.  dynamic LDS: keys[P] (u64) | cell[root*root] (u32) | scan[NT] (int) | misc.  NT = 256 threads, or 1024 when
// thousands of matches per pair are sorted (N = 4000 / 8000: the launch has only `pairs` workgroups)
template <int NT>
__global__ __launch_bounds__(NT) void k_filter(const vis_keypoint* __restrict__ kps, const int32_t* __restrict__ nkp, int kcap,
                                                const int32_t* __restrict__ pair_q, const int32_t* __restrict__ pair_t,
                                                const uint32_t* __restrict__ knn12, const uint32_t* __restrict__ knn21,
                                                double ratio, int sym_mode, int root,
                                                const float* __restrict__ hf, const float* __restrict__ wf,
                                                vis_dmatch* __restrict__ sym_out, int32_t* __restrict__ nsym_out,
                                                vis_dmatch* __restrict__ good_out, int32_t* __restrict__ ngood_out,
                                                float* __restrict__ p1, float* __restrict__ p2, int keys_cap, int pose_mcap, int pose_sym) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    uint32_t* cell = reinterpret_cast<uint32_t*>(smem + (size_t)keys_cap * 8);
    int* scan = reinterpret_cast<int*>(cell + root * root);
    int* misc = scan + NT;
    const int tid = threadIdx.x, pair = blockIdx.x;
    const int rq = pair_q[pair], rt = pair_t[pair];
    const int ncell = root * root;
    if (rq < 0 || rt < 0) { if (tid == 0) { nsym_out[pair] = 0; ngood_out[pair] = 0; } return; }
    const int n1 = min(nkp[rq], kcap), n2 = min(nkp[rt], kcap);
    const vis_keypoint* K1 = kps + (size_t)rq * kcap;
    const vis_keypoint* K2 = kps + (size_t)rt * kcap;
    const uint32_t* A = knn12 + (size_t)pair * kcap * 2;
    const uint32_t* Bk = knn21 + (size_t)pair * kcap * 2;
    vis_dmatch* sym = sym_out + (size_t)pair * kcap;
    // ---- computeSymMatches: ordered compaction over q (chunk per thread + block scan)
    const int chunk = (n1 + NT - 1) / NT;
    const int qb = tid * chunk, qe = min(n1, qb + chunk);
    int cnt = 0;
    for (int q = qb; q < qe; q++) {
        const uint32_t a0 = A[2 * q], a1 = A[2 * q + 1];
        bool ok = ratio_survives(a0, a1, ratio);
        if (ok) {
            const int t = a0 & 0xFFFF;
            ok = t < n2;
            if (ok) {
                const uint32_t b0 = Bk[2 * t], b1 = Bk[2 * t + 1];
                ok = (b0 != 0xFFFFFFFFu) && (int)(b0 & 0xFFFF) == q;
                if (ok && sym_mode == VIS_SYM_INTENDED) ok = ratio_survives(b0, b1, ratio);
            }
        }
        cnt += ok ? 1 : 0;
    }
    scan[tid] = cnt;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int i = 0; i < NT; i++) { int c = scan[i]; scan[i] = acc; acc += c; } misc[0] = acc; }
    __syncthreads();
    const int nsym = misc[0];
    int pos = scan[tid];
    for (int q = qb; q < qe; q++) {
        const uint32_t a0 = A[2 * q], a1 = A[2 * q + 1];
        bool ok = ratio_survives(a0, a1, ratio);
        if (ok) {
            const int t = a0 & 0xFFFF;
            ok = t < n2;
            if (ok) {
                const uint32_t b0 = Bk[2 * t], b1 = Bk[2 * t + 1];
                ok = (b0 != 0xFFFFFFFFu) && (int)(b0 & 0xFFFF) == q;
                if (ok && sym_mode == VIS_SYM_INTENDED) ok = ratio_survives(b0, b1, ratio);
            }
            if (ok) {
                vis_dmatch m; m.queryIdx = q; m.trainIdx = t; m.imgIdx = -1; m.distance = (float)(a0 >> 16);
                sym[pos] = m;
                // sortMatches key: y ascending, ties keep symmetric-match order (stable)
                keys[pos] = ((uint64_t)fmap_f(K1[q].y) << 32) | (uint32_t)pos;
                pos++;
            }
        }
    }
    int P2 = 2; while (P2 < nsym) P2 <<= 1;
    __syncthreads();
    if (pose_sym) {                                                // VIS_POSE_SYM: the pose stage takes every symmetric match
        float* q1 = p1 + (size_t)pair * pose_mcap * 2;
        float* q2 = p2 + (size_t)pair * pose_mcap * 2;
        for (int i = tid; i < min(nsym, pose_mcap); i += NT) {
            const vis_dmatch m = sym[i];
            q1[2 * i] = K1[m.queryIdx].x; q1[2 * i + 1] = K1[m.queryIdx].y;
            q2[2 * i] = K2[m.trainIdx].x; q2[2 * i + 1] = K2[m.trainIdx].y;
        }
    }
    for (int i = nsym + tid; i < P2; i += NT) keys[i] = ~0ull;
    for (int i = tid; i < ncell; i += NT) cell[i] = 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P2; i += NT) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t a = keys[i], b = keys[ixj];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // ---- bestMatchesFilter: band = first j with y <= hf[j]; column = steps until x <= wf[i] (clamped);
    // per cell the strictly smallest distance, first in sorted order wins ties -> min of (dist<<16 | sorted pos)
    for (int s = tid; s < nsym; s += NT) {
        const vis_dmatch m = sym[(uint32_t)keys[s]];
        const float y = K1[m.queryIdx].y, x = K1[m.queryIdx].x;
        int band = -1;
        for (int j = 0; j < root; j++) if (y <= hf[j]) { band = j; break; }
        if (band < 0) continue;
        int col = 0;
        while (col < root - 1 && x > wf[col]) col++;
        atomicMin(&cell[band * root + col], ((uint32_t)m.distance << 16) | (uint32_t)s);
    }
    __syncthreads();
    // ordered compaction of the occupied cells (band-major, column ascending)
    if (tid == 0) {
        int n = 0;
        vis_dmatch* good = good_out + (size_t)pair * ncell;
        float* q1 = p1 + (size_t)pair * ncell * 2;
        float* q2 = p2 + (size_t)pair * ncell * 2;
        for (int c = 0; c < ncell; c++) {
            const uint32_t v = cell[c];
            if (v == 0xFFFFFFFFu) continue;
            const vis_dmatch m = sym[(uint32_t)keys[v & 0xFFFF]];
            good[n] = m;
            if (!pose_sym) {
                q1[2 * n] = K1[m.queryIdx].x; q1[2 * n + 1] = K1[m.queryIdx].y;     // getGoodMatches
                q2[2 * n] = K2[m.trainIdx].x; q2[2 * n + 1] = K2[m.trainIdx].y;
            }
            n++;
        }
        ngood_out[pair] = n;
        nsym_out[pair] = nsym;
    }
}

int launch_expand(vis_ctx* ctx, Plan* pl, int rec_first, int rec_count) {
    if (rec_count <= 0 || !pl->d_descx) return VIS_OK;
    const long long threads = (long long)rec_count * pl->kcap * 8;
    hipLaunchKernelGGL(k_expand, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, pl->d_desc, pl->d_nkp, pl->kcap,
                       pl->d_descx, rec_first, rec_count);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}

int launch_match(vis_ctx* ctx, Plan* pl, int npairs) {
    if (npairs <= 0) return VIS_OK;
    // A/B knobs exist only in the diagnostic build (make EXTRA=-DVIS_AB_KNOBS): the shipped library reads no environment variable
#ifdef VIS_AB_KNOBS
    static const bool force_popcount = getenv("VIS_KNN_POPCOUNT") != nullptr;
    static const int nc = getenv("VIS_KNN_NC") ? atoi(getenv("VIS_KNN_NC")) : 2;
#else
    const bool force_popcount = false;
    const int nc = 2;
#endif
    if (pl->d_descx && pl->kcap <= 16384 && !force_popcount) {
        const int per_wg = 128 * (nc == 1 ? 1 : 2);
        const int nchunks = (pl->kcap + per_wg - 1) / per_wg;
        dim3 grid(8 * ((npairs + 7) / 8) * 2 * nchunks);
        if (nc == 1)
            hipLaunchKernelGGL(k_knn_mfma<1>, grid, dim3(256), 0, ctx->stream, pl->d_descx, pl->d_nkp, pl->kcap,
                               pl->d_pair_q, pl->d_pair_t, pl->d_knn12, pl->d_knn21, npairs, nchunks);
        else
            hipLaunchKernelGGL(k_knn_mfma<2>, grid, dim3(256), 0, ctx->stream, pl->d_descx, pl->d_nkp, pl->kcap,
                               pl->d_pair_q, pl->d_pair_t, pl->d_knn12, pl->d_knn21, npairs, nchunks);
        HIPCHK(ctx, hipGetLastError());
        return VIS_OK;
    }
    // small problems: one wave per block so a single pair still spreads over many CUs
    const int bs = (npairs * ((pl->kcap + 255) / 256) * 2 >= 512) ? 256 : 64;
    dim3 grid((pl->kcap + 2 * bs - 1) / (2 * bs), npairs, 2);
    hipLaunchKernelGGL(k_knn2, grid, dim3(bs), 0, ctx->stream, pl->d_desc, pl->d_nkp, pl->kcap,
                       pl->d_pair_q, pl->d_pair_t, pl->d_knn12, pl->d_knn21);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}

int launch_filter(vis_ctx* ctx, Plan* pl, int npairs) {
    if (npairs <= 0) return VIS_OK;
    int keys_cap = 2; while (keys_cap < pl->kcap) keys_cap <<= 1;
    const int nt = pl->kcap > 2048 ? 1024 : 256;
    const size_t lds = (size_t)keys_cap * 8 + (size_t)pl->root * pl->root * 4 + (size_t)nt * 4 + 16;
    if (lds > 160 * 1024) { ctx->err = "the match filters sort up to 16384 keypoints per frame in LDS: keypoint capacity " + std::to_string(pl->kcap) + " is beyond that"; return VIS_E_CAPACITY; }
    auto kern = nt == 1024 ? k_filter<1024> : k_filter<256>;
    if (lds > 65536) HIPCHK(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(npairs), dim3(nt), lds, ctx->stream, pl->d_kps, pl->d_nkp, pl->kcap,
                       pl->d_pair_q, pl->d_pair_t, pl->d_knn12, pl->d_knn21, (double)ctx->p.ratio, ctx->p.sym_mode,
                       pl->root, pl->d_hf, pl->d_wf, pl->d_sym, pl->d_nsym, pl->d_good, pl->d_ngood,
                       pl->d_p1, pl->d_p2, keys_cap, pl->pose_mcap ? pl->pose_mcap : pl->root * pl->root,
                       (pl->pose_mcap && ctx->p.pose_input == VIS_POSE_SYM) ? 1 : 0);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}
