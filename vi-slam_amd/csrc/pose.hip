// pose.hip -- batched essential-matrix RANSAC (5-point + Sampson), recoverPose (cheirality vote) and
// F2FRansac on gfx950, FP64 VALU.
//
// Replaces the calib3d calls of VISystem::EstimatePoseFeaturesRansac
//   findEssentialMat(p1,p2,focal,pp,RANSAC,0.999,1.0)   /root/reference/src/VISystem.cpp:1679-1680
//   recoverPose(E,p1,p2,R,t,focal,pp)                   /root/reference/src/VISystem.cpp:1701
// and the reference's own VISystem::F2FRansac            /root/reference/src/VISystem.cpp:612-769.
//
// Parallelisation: RANSAC is sequential only in (a) its cv::RNG sample stream and (b) the
// "first strictly better model wins / adaptive iteration count" scan.  Both are O(iters) scalar work;
// everything expensive (minimal solver, M Sampson errors per model) is independent per hypothesis:
//   k_pose_prep    block/pair : normalise points, copy (or replay) the cv::RNG sample table
//   k_ransac_hyp   lane/hypothesis : minimal solver up to the degree-10 polynomial -> hypothesis record
//   k_hyp_roots    16 lanes/hypothesis : real roots, one bisection interval per lane (a handful of pairs: latency)
//   k_hyp_roots_packed  wave/64 hypotheses : the same, the intervals of a level packed over the lanes (batches: throughput)
//   k_hyp_models   lane/hypothesis : back-substitution, <= 10 candidate E per hypothesis
//   k_hyp_score    256 threads/16 hypotheses : inlier counts of every E (single precision inside error radii, double otherwise)
//   k_ransac_scan  wave/pair : replays RANSACPointSetRegistrator::run's accept/update rule in order
//   k_pose_final   block/pair : inlier mask of the winner, SVD, 4x M DLT triangulations, cheirality vote
// The first 16 hypotheses of every pair are evaluated and scanned first; later chunks run only for the pairs whose
// adaptive iteration count is still above 16 (device work list) -- the rare case on good matches.
// Algorithm and operation order mirror oracle/pose.cpp; compared with a stated tolerance
// (tests/test_pose_gpu.py), not bit-exactly.
#include "vis_internal.h"
#include <type_traits>
#include <cfloat>
#include <cstdlib>

#define DEV __device__ __forceinline__

struct PoseTables { int8_t q_of[4][4]; int8_t c_of[10][4]; };
constexpr int QE[10][3] = {{2,0,0},{1,1,0},{1,0,1},{0,2,0},{0,1,1},{0,0,2},{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
constexpr int CE[20][3] = {{3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
                           {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0}};
constexpr PoseTables make_tables() {
    PoseTables t{};
    const int ve[4][3] = {{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
        for (int m = 0; m < 10; m++)
            if (QE[m][0] == ve[i][0] + ve[j][0] && QE[m][1] == ve[i][1] + ve[j][1] && QE[m][2] == ve[i][2] + ve[j][2]) t.q_of[i][j] = (int8_t)m;
    for (int m = 0; m < 10; m++) for (int v = 0; v < 4; v++)
        for (int c = 0; c < 20; c++)
            if (CE[c][0] == QE[m][0] + ve[v][0] && CE[c][1] == QE[m][1] + ve[v][1] && CE[c][2] == QE[m][2] + ve[v][2]) t.c_of[m][v] = (int8_t)c;
    return t;
}
__device__ const PoseTables g_tb = make_tables();

DEV void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
DEV double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// cyclic Jacobi eigen-decomposition of a symmetric N x N matrix (oracle/pose.cpp jacobi_eig: same rotations in the same order).
// N is a template parameter and every index a constant: the matrices stay in registers (the run-time-n form kept them in
// scratch memory and spent more instructions on addresses than on the rotations).
template <int N>
__device__ __forceinline__ void jacobi_eig(double (&A)[N * N], double (&V)[N * N], bool* zero_theta_last = nullptr) {
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) V[i * N + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int j = i + 1; j < N; j++) off += A[i * N + j] * A[i * N + j];
        if (off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < N; p++)
#pragma unroll
            for (int q = p + 1; q < N; q++) {
                const double apq = A[p * N + q];
                if (fabs(apq) < 1e-300) continue;
                const double app = A[p * N + p], aqq = A[q * N + q];
                const double theta = (aqq - app) / (2.0 * apq);
                // (theta == +-0 takes t = +1 whatever the sign of apq: the one place where negating row / column N-1 of A does not
                // simply negate the matching entries of everything that follows -- see cheirality_pair)
                if (zero_theta_last && q == N - 1 && theta == 0.0) *zero_theta_last = true;
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double akp = A[k * N + p], akq = A[k * N + q];
                    A[k * N + p] = c * akp - s * akq;
                    A[k * N + q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double apk = A[p * N + k], aqk = A[q * N + k];
                    A[p * N + k] = c * apk - s * aqk;
                    A[q * N + k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double vkp = V[k * N + p], vkq = V[k * N + q];
                    V[k * N + p] = c * vkp - s * vkq;
                    V[k * N + q] = s * vkp + c * vkq;
                }
            }
    }
}

DEV void mul_ll(const double* a, const double* b, double* q) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) q[g_tb.q_of[i][j]] += a[i] * b[j];
}
DEV void mul_ql(const double* q, const double* l, double* c, double sgn) {
    for (int m = 0; m < 10; m++) for (int v = 0; v < 4; v++) c[g_tb.c_of[m][v]] += sgn * (q[m] * l[v]);
}
DEV double poly_eval(const double* c, int deg, double x) {
    double r = c[deg];
    for (int i = deg - 1; i >= 0; i--) r = r * x + c[i];
    return r;
}

#define BISECT_INNER 40
#define BISECT_FINAL 200

DEV void pmul(const double* a, int da, const double* b, int db, double* o) {
    for (int i = 0; i <= da + db; i++) o[i] = 0;
    for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) o[i + j] += a[i] * b[j];
}

// ---- compile-time monomial tables for the unrolled solver (same tables as g_tb)
struct CT {
    static constexpr PoseTables T = make_tables();
};

// lin*lin -> quad, quad*lin -> cubic with compile-time slots (register arrays)
DEV void mul_ll_u(const double (&a)[4], const double (&b)[4], double (&q)[10]) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) q[CT::T.q_of[i][j]] += a[i] * b[j];
}
DEV void mul_ql_u(const double (&q)[10], const double (&l)[4], double (&c)[20], double sgn) {
#pragma unroll
    for (int m = 0; m < 10; m++)
#pragma unroll
        for (int v = 0; v < 4; v++) c[CT::T.c_of[m][v]] += sgn * (q[m] * l[v]);
}

// LDS layout (one wave per block): FOUR lanes per hypothesis, 16 hypotheses per wave; element-major, hypothesis-minor.
// 116 doubles per hypothesis = 14.5 KB per wave (a four-row stage + the null-space basis; the 10 x 20 system itself lives in the
// quad's registers): ten waves per CU as far as LDS goes (the lane-per-hypothesis layout needed 118 KB: one).
#define QH 16
#define LS(s_, c) ldsM[((s_) * 20 + (c)) * QH + hs]                 // stage slot s_ (four rows of 20) of the row -> column exchange
#define LR(r, c) ldsM[(((r) - 4) * 10 + ((c) - 10)) * QH + hs]    // the same memory afterwards: right-hand block of rows 4..9
#define LB(j, i) ldsB[((j) * 9 + (i)) * QH + hs]
#define HYP_STAGE 80
#define HYP_LDS_BYTES ((HYP_STAGE + 36) * QH * 8)
// hypothesis record (doubles, element-major over all (pair, iteration) slots): det polynomial c[0..10], the three
// B(z) row polynomials, the null-space basis, the real roots; then two int32 planes: flag, number of roots
#define HR_BX 11
#define HR_BY 23
#define HR_B1 35
#define HR_LB 50
#define HR_ROOTS 86
#define HR_DOUBLES 96
// one wave per block: LDS accesses of a wave execute in order, only the compiler has to be kept from moving them
#define HYP_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

DEV void load_El(const double* ldsB, int hs, int r, int c, double (&l)[4]) {
#pragma unroll
    for (int v = 0; v < 4; v++) l[v] = LB(v, 3 * r + c);
}

DEV int sampson_inlier(const double* E, double x1, double y1, double x2, double y2, float t) {
    const double Ex0 = (E[0] * x1 + E[1] * y1) + E[2];
    const double Ex1 = (E[3] * x1 + E[4] * y1) + E[5];
    const double Ex2 = (E[6] * x1 + E[7] * y1) + E[8];
    const double Et0 = (E[0] * x2 + E[3] * y2) + E[6];
    const double Et1 = (E[1] * x2 + E[4] * y2) + E[7];
    const double x2tEx1 = (x2 * Ex0 + y2 * Ex1) + Ex2;
    const double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    const float err = (float)(x2tEx1 * x2tEx1 / (((a + b) + c) + d));
    return err <= t ? 1 : 0;
}

// cv::RNG
struct CvRng {
    unsigned long long state;
    DEV unsigned next() { state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32); return (unsigned)state; }
    DEV int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
    // next() % m for a divisor that stays the same over thousands of draws: q = floor(x * ceil(2^64 / m) / 2^64) is the exact
    // quotient for every 32-bit x (Lemire), so the remainder needs two multiplies instead of a hardware-less 32-bit division
    DEV unsigned next_mod(unsigned m, unsigned long long magic) { const unsigned x = next(); return x - (unsigned)__umul64hi(magic, (unsigned long long)x) * m; }
};

struct PoseParams {
    double fx_inv, cx, cy, thr, prob;
    unsigned long long seed;
    int max_iters, adaptive, mcap;
    const int32_t* sample_table;     // host-built cv::RNG sample tables for M in [6, table_max_m] (may be null)
    int table_max_m;
};

// rstate: [0] niters, [1] maxGood, [2] best hypothesis index (-1 none), [3] best model, [4] next iteration to scan,
//         [5] special (1 = M==5 shortcut, 2 = M<5), [6] M, [7] iterations run, [8] candidate models scored (all chunks),
//         [9..12] cheirality votes of the four (R, t) candidates, [14] finished workgroups (k_pose_final)
#define RS VIS_RSTATE_WORDS

// getSubset (ptsetreg.cpp): 5 distinct indices in [0,M), redraw on duplicates, from the running cv::RNG
DEV void draw_subset(CvRng& rng, int M, int* idx) {
    const unsigned long long magic = 0xFFFFFFFFFFFFFFFFull / (unsigned)M + 1ull;         // M >= 6 here
    for (int i = 0; i < 5; i++) {
        for (;;) {
            const int v = idx[i] = (int)rng.next_mod((unsigned)M, magic);
            int j = 0;
            for (; j < i; j++) if (v == idx[j]) break;
            if (j == i) break;
        }
    }
}

__global__ __launch_bounds__(256) void k_pose_prep(PoseParams P, const float* __restrict__ p1, const float* __restrict__ p2,
                                                   const int32_t* __restrict__ npts, double* __restrict__ n1, double* __restrict__ n2,
                                                   int32_t* __restrict__ samples, int32_t* __restrict__ rstate, int32_t* __restrict__ worklist) {
    const int pair = blockIdx.x, tid = threadIdx.x;
    // the work list starts empty (here instead of a hipMemsetAsync in front of the RANSAC kernels: one launch less on the pose stream,
    // and the runtime's fill kernel is not an ordinary launch -- see vis_batch_results_async on what its copies cost the pipeline)
    if (worklist && pair == 0 && tid == 0) worklist[0] = 0;
    const int M = min(npts[pair], P.mcap);
    const float* a = p1 + (size_t)pair * P.mcap * 2;
    const float* b = p2 + (size_t)pair * P.mcap * 2;
    double* o1 = n1 + (size_t)pair * P.mcap * 2;
    double* o2 = n2 + (size_t)pair * P.mcap * 2;
    for (int i = tid; i < M; i += 256) {
        o1[2 * i] = ((double)a[2 * i] - P.cx) * P.fx_inv; o1[2 * i + 1] = ((double)a[2 * i + 1] - P.cy) * P.fx_inv;
        o2[2 * i] = ((double)b[2 * i] - P.cx) * P.fx_inv; o2[2 * i + 1] = ((double)b[2 * i + 1] - P.cy) * P.fx_inv;
    }
    int32_t* rs = rstate + (size_t)pair * RS;
    int32_t* sm = samples + (size_t)pair * P.max_iters * 5;
    if (tid == 0) {
        rs[1] = 0; rs[2] = -1; rs[3] = 0; rs[4] = 0; rs[6] = M; rs[7] = 0; rs[8] = 0;
        for (int k = 9; k < 15; k++) rs[k] = 0;
        if (M < 5) { rs[0] = 0; rs[5] = 2; }
        else if (M == 5) { rs[0] = 1; rs[5] = 1; for (int k = 0; k < 5; k++) sm[k] = k; }
        else { rs[0] = max(P.max_iters, 1); rs[5] = 0; }
    }
    if (M > 5) {
        if (P.sample_table && M <= P.table_max_m) {              // table lookup: parallel copy
            const int32_t* src = P.sample_table + (size_t)(M - 6) * P.max_iters * 5;
            for (int i = tid; i < P.max_iters * 5; i += 256) sm[i] = src[i];
        } else if (tid == 0) {                                   // sequential replay of the RNG stream
            CvRng rng; rng.state = P.seed ? P.seed : 0xffffffffULL;
            for (int it = 0; it < P.max_iters; it++) {
                int idx[5]; draw_subset(rng, M, idx);
                for (int k = 0; k < 5; k++) sm[5 * it + k] = idx[k];
            }
        }
    }
}

// the value lane SRC of every quad holds, in all four lanes of the quad (two DPP moves per double)
template <int SRC>
DEV double quad_bcast(double x) {
    constexpr int ctrl = SRC * 0x55;                               // quad_perm [SRC, SRC, SRC, SRC]
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, ctrl, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), ctrl, 0xF, 0xF, false);
    return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}

// One Gauss-Jordan step (column COL) of the quad's 10 x 20 system: M[r][cc] = element (r, 4 cc + q) in lane q.  The pivot column
// comes from its owner lane by quad broadcast; pivot search, reciprocal and multipliers are replicated in the four lanes; a lane
// updates its own columns c >= COL.  Per element exactly the operations of oracle/pose.cpp five_point(): rows COL and piv are
// exchanged, the pivot row is scaled by 1 / pivot, every other row with a non-zero multiplier subtracts multiplier * pivot row.
template <int COL>
DEV void gj_step(double (&M)[10][5], int q, bool& ok) {
    double v[10];
#pragma unroll
    for (int r = 0; r < 10; r++) v[r] = quad_bcast<(COL & 3)>(M[r][COL >> 2]);
    int piv = COL; double best = fabs(v[COL]);
#pragma unroll
    for (int r = COL + 1; r < 10; r++) { const double a = fabs(v[r]); if (a > best) { best = a; piv = r; } }
    if (best < 1e-300) ok = false;
    const double pv = v[COL];                                      // after the row exchange row `piv` holds the old row COL
    double vp = v[COL];
#pragma unroll
    for (int r = COL + 1; r < 10; r++) vp = (r == piv) ? v[r] : vp;
    const double inv = 1.0 / vp;
#pragma unroll
    for (int cc = 0; cc < 5; cc++) {
        if (4 * cc + 3 < COL) continue;                            // columns left of the pivot are finished
        if (4 * cc + q >= COL) {
            const double t = M[COL][cc];
            double pr = t;
#pragma unroll
            for (int r = COL + 1; r < 10; r++)
                if (r == piv) { pr = M[r][cc]; M[r][cc] = t; }
            pr *= inv;
            M[COL][cc] = pr;
#pragma unroll
            for (int r = 0; r < 10; r++) {
                if (r == COL) continue;
                const double f = (r == piv) ? pv : v[r];
                if (f != 0.0) M[r][cc] -= f * pr;
            }
        }
    }
}

// One wave per block, FOUR lanes per hypothesis (quad lane q = lane & 3, hypothesis slot hs = lane >> 2), 16 consecutive
// hypotheses of ONE frame pair per wave.  Every matrix element is produced by exactly the operation sequence of
// oracle/pose.cpp five_point() -- the quad only decides WHICH lane runs a sequence:
//   Householder QR of the 9 x 5 epipolar system     replicated in the four lanes (registers, ~600 flops)
//   null-space basis vector j                         lane q = j
//   constraint row 0 (det E)                          lane 0;      rows 1 + 3 i + j (j = 0..2)    lane q = i + 1
//   Gauss-Jordan, partial pivoting                    every lane reads column `col` (pivot search + the 10 multipliers), lane q
//                                                     updates the columns c = q (mod 4) of all rows
//   B(z), det B(z) (degree 10), record header         lane 0;      the 36 basis doubles of the record    lanes 1..3
// hypothesis record layout and everything downstream (k_hyp_roots, k_hyp_score) are unchanged.
// The first launch covers hypotheses [0, first) of every pair (grid.y = pair); later chunks run only for the pairs the
// first scan put on the device work list: a fixed grid walks (pair, 16-hypothesis chunk) items.
__device__ __forceinline__ void ransac_hyp_quad(const PoseParams& P, int pair_raw, int hbase, int h_end, int npairs,
                                             const double* __restrict__ n1, const double* __restrict__ n2,
                                             const int32_t* __restrict__ samples, const int32_t* __restrict__ rstate,
                                             double* __restrict__ hyp, size_t S,
                                             double* ldsM, double* ldsB) {
    const int lane = threadIdx.x, hs = lane >> 2, q = lane & 3;
    const int pair = min(pair_raw, npairs - 1);
    const int h = hbase + hs;
    const int32_t* rs = rstate + (size_t)pair * RS;
    const int niters = rs[0];
    const bool active = pair_raw < npairs && h < niters && h < h_end && h < max(P.max_iters, 1);
    if (!__any(active)) return;                                    // adaptive stop already below this wave
    const double* pa = n1 + (size_t)pair * P.mcap * 2;
    const double* pb = n2 + (size_t)pair * P.mcap * 2;
    const int hh = active ? h : 0;                                 // inactive quads redo hypothesis 0 (results dropped)
    const int32_t* sm = samples + ((size_t)pair * P.max_iters + hh) * 5;
    // ---- epipolar system A = Q^T (9 x 5)
    double A[9][5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        // inactive lanes (their pair has no work, e.g. M < 5 or the adaptive bound is already reached) share the
        // wave with active ones: they must not touch the (possibly never written) sample table
        const int id = active ? sm[i] : 0;
        const double x1 = pa[2 * id], y1 = pa[2 * id + 1], x2 = pb[2 * id], y2 = pb[2 * id + 1];
        A[0][i] = x2 * x1; A[1][i] = x2 * y1; A[2][i] = x2;
        A[3][i] = y2 * x1; A[4][i] = y2 * y1; A[5][i] = y2;
        A[6][i] = x1; A[7][i] = y1; A[8][i] = 1.0;
    }
    // ---- Householder QR, reflectors kept (vs[k][i] for i >= k)
    double vs[5][9]; double betas[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        double nrm = 0;
#pragma unroll
        for (int i = k; i < 9; i++) nrm += A[i][k] * A[i][k];
        nrm = sqrt(nrm);
#pragma unroll
        for (int i = 0; i < 9; i++) vs[k][i] = 0;
        double beta = 0;
        if (!(nrm < 1e-300)) {
            const double alpha = A[k][k] >= 0 ? -nrm : nrm;
#pragma unroll
            for (int i = k; i < 9; i++) vs[k][i] = A[i][k];
            vs[k][k] -= alpha;
            double vn = 0;
#pragma unroll
            for (int i = k; i < 9; i++) vn += vs[k][i] * vs[k][i];
            if (!(vn < 1e-300)) {
                beta = 2.0 / vn;
#pragma unroll
                for (int j = k; j < 5; j++) {
                    double d = 0;
#pragma unroll
                    for (int i = k; i < 9; i++) d += vs[k][i] * A[i][j];
                    d *= beta;
#pragma unroll
                    for (int i = k; i < 9; i++) A[i][j] -= d * vs[k][i];
                }
            }
        }
        betas[k] = beta;
    }
    // ---- null-space basis X,Y,Z,W = columns 5..8 of H0..H4 -> LDS: lane q builds vector j = q
    {
        double e[9];
#pragma unroll
        for (int i = 0; i < 9; i++) e[i] = (i == 5 + q) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 4; k >= 0; k--) {
            double d = 0;
#pragma unroll
            for (int i = k; i < 9; i++) d += vs[k][i] * e[i];
            d *= betas[k];
#pragma unroll
            for (int i = k; i < 9; i++) e[i] -= d * vs[k][i];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) LB(q, i) = e[i];
    }
    HYP_SYNC();
    // ---- constraint rows.  Row 0 (det E) is built by lane 0, rows 1 + 3 i + j of (E E^T - 0.5 tr(E E^T) I) E by lane q = i + 1;
    // the Gauss-Jordan below wants COLUMNS c = q (mod 4) of all ten rows in lane q.  The exchange goes through a four-row LDS
    // stage in three rounds (round j: rows 1 + j, 4 + j, 7 + j, and row 0 in the first), so a wave holds 14.5 KB of LDS instead of
    // the 30 KB of the whole 10 x 20 system: ten waves per CU fit instead of five.
    double Mq[10][5];
    double EEt[3][10];
    if (q != 0) {
        // lane q owns i = q - 1 (all index arithmetic on i goes through LDS)
        const int i = q - 1;
        double tr[10];
        {
            double dg[3][10];
#pragma unroll
            for (int ii = 0; ii < 3; ii++) {
#pragma unroll
                for (int m = 0; m < 10; m++) dg[ii][m] = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) { double l1[4]; load_El(ldsB, hs, ii, k, l1); mul_ll_u(l1, l1, dg[ii]); }
            }
#pragma unroll
            for (int m = 0; m < 10; m++) tr[m] = 0.5 * ((dg[0][m] + dg[1][m]) + dg[2][m]);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
#pragma unroll
            for (int m = 0; m < 10; m++) EEt[j][m] = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { double l1[4], l2[4]; load_El(ldsB, hs, i, k, l1); load_El(ldsB, hs, j, k, l2); mul_ll_u(l1, l2, EEt[j]); }
        }
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int m = 0; m < 10; m++) EEt[j][m] = (j == i) ? EEt[j][m] - tr[m] : EEt[j][m];
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (q == 0) {
            if (j == 0) {
                double row[20];
#pragma unroll
                for (int c = 0; c < 20; c++) row[c] = 0;
                constexpr int ta[3] = {0, 1, 2}, tb[3] = {1, 0, 0}, tc[3] = {2, 2, 1}, td[3] = {2, 2, 1}, te[3] = {1, 0, 0};
                constexpr double sg[3] = {1.0, -1.0, 1.0};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double qq[10], q2[10], l1[4], l2[4];
#pragma unroll
                    for (int i = 0; i < 10; i++) { qq[i] = 0; q2[i] = 0; }
                    load_El(ldsB, hs, 1, tb[k], l1); load_El(ldsB, hs, 2, tc[k], l2); mul_ll_u(l1, l2, qq);
                    load_El(ldsB, hs, 1, td[k], l1); load_El(ldsB, hs, 2, te[k], l2); mul_ll_u(l1, l2, q2);
#pragma unroll
                    for (int i = 0; i < 10; i++) qq[i] -= q2[i];
                    load_El(ldsB, hs, 0, ta[k], l1);
                    mul_ql_u(qq, l1, row, sg[k]);
                }
#pragma unroll
                for (int c = 0; c < 20; c++) LS(0, c) = row[c];
            }
        } else {
            double row[20];
#pragma unroll
            for (int c = 0; c < 20; c++) row[c] = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) { double l1[4]; load_El(ldsB, hs, k, j, l1); mul_ql_u(EEt[k], l1, row, 1.0); }
#pragma unroll
            for (int c = 0; c < 20; c++) LS(q, c) = row[c];
        }
        HYP_SYNC();
#pragma unroll
        for (int cc = 0; cc < 5; cc++) {
            if (j == 0) Mq[0][cc] = LS(0, 4 * cc + q);
            Mq[1 + j][cc] = LS(1, 4 * cc + q); Mq[4 + j][cc] = LS(2, 4 * cc + q); Mq[7 + j][cc] = LS(3, 4 * cc + q);
        }
        HYP_SYNC();
    }
    // ---- Gauss-Jordan with partial pivoting on the left 10 x 10 block, IN REGISTERS (gj_step<col>); only the right-hand block of
    // rows 4..9 -- what B(z) is built from -- goes back to LDS.  (With the matrix in LDS every element update was a dependent LDS
    // round trip of a kernel that ran at one wave per SIMD: 1.2 of its 2.0 ms per 1.02 M hypotheses.)
    bool ok = true;
    gj_step<0>(Mq, q, ok); gj_step<1>(Mq, q, ok); gj_step<2>(Mq, q, ok); gj_step<3>(Mq, q, ok); gj_step<4>(Mq, q, ok);
    gj_step<5>(Mq, q, ok); gj_step<6>(Mq, q, ok); gj_step<7>(Mq, q, ok); gj_step<8>(Mq, q, ok); gj_step<9>(Mq, q, ok);
#pragma unroll
    for (int r = 4; r < 10; r++)
#pragma unroll
        for (int cc = 2; cc < 5; cc++)
            if (4 * cc + q >= 10) LR(r, 4 * cc + q) = Mq[r][cc];
    HYP_SYNC();
    // ---- the record's 36 null-space doubles: lanes 1..3, 12 each (element-major, slot-minor: a quad lane's 16 hypotheses are
    // 16 consecutive slots)
    const size_t slot = (size_t)pair * P.max_iters + (active ? h : 0);
    if (active && q != 0) {
        double* rec = hyp + slot;
#pragma unroll
        for (int e = 0; e < 12; e++) {
            const int idx = (q - 1) * 12 + e;                      // j * 9 + i
            rec[(size_t)(HR_LB + idx) * S] = ldsB[idx * QH + hs];
        }
    }
    if (q == 0) {
    // ---- B(z) from rows (4,5),(6,7),(8,9)
    double Bx[3][4], By[3][4], B1[3][5];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double a[10], b[10];
#pragma unroll
        for (int c = 0; c < 10; c++) { a[c] = LR(4 + 2 * i, 10 + c); b[c] = LR(5 + 2 * i, 10 + c); }
        Bx[i][0] = a[2]; Bx[i][1] = a[1] - b[2]; Bx[i][2] = a[0] - b[1]; Bx[i][3] = -b[0];
        By[i][0] = a[5]; By[i][1] = a[4] - b[5]; By[i][2] = a[3] - b[4]; By[i][3] = -b[3];
        B1[i][0] = a[9]; B1[i][1] = a[8] - b[9]; B1[i][2] = a[7] - b[8]; B1[i][3] = a[6] - b[7]; B1[i][4] = -b[6];
    }
    // ---- det B(z): degree-10 polynomial (same product/summation order as the oracle)
    double c10[11];
#pragma unroll
    for (int i = 0; i <= 10; i++) c10[i] = 0;
    {
        double t1[8], t2[8], m[8], o[11];
#define PMUL(a, da, b, db, o_) do { _Pragma("unroll") for (int i_ = 0; i_ <= (da) + (db); i_++) (o_)[i_] = 0; \
            _Pragma("unroll") for (int i_ = 0; i_ <= (da); i_++) _Pragma("unroll") for (int j_ = 0; j_ <= (db); j_++) (o_)[i_ + j_] += (a)[i_] * (b)[j_]; } while (0)
        PMUL(By[1], 3, B1[2], 4, t1); PMUL(B1[1], 4, By[2], 3, t2);
#pragma unroll
        for (int i = 0; i <= 7; i++) m[i] = t1[i] - t2[i];
        PMUL(Bx[0], 3, m, 7, o);
#pragma unroll
        for (int i = 0; i <= 10; i++) c10[i] += o[i];
        PMUL(Bx[1], 3, B1[2], 4, t1); PMUL(B1[1], 4, Bx[2], 3, t2);
#pragma unroll
        for (int i = 0; i <= 7; i++) m[i] = t1[i] - t2[i];
        PMUL(By[0], 3, m, 7, o);
#pragma unroll
        for (int i = 0; i <= 10; i++) c10[i] -= o[i];
        double u1[7], u2[7], mm[7];
        PMUL(Bx[1], 3, By[2], 3, u1); PMUL(By[1], 3, Bx[2], 3, u2);
#pragma unroll
        for (int i = 0; i <= 6; i++) mm[i] = u1[i] - u2[i];
        PMUL(B1[0], 4, mm, 6, o);
#pragma unroll
        for (int i = 0; i <= 10; i++) c10[i] += o[i];
#undef PMUL
    }
    // ---- hypothesis record -> global memory (element-major, slot-minor)
    if (active) {
        double* rec = hyp + slot;
        double mx = 0;
#pragma unroll
        for (int i = 0; i <= 10; i++) mx = fmax(mx, fabs(c10[i]));
        int flag = 0;                                              // 0: no model, 1: degree 10, 2: leading coefficient lost
        if (!(mx == 0 || !ok)) {
            double c[11];
#pragma unroll
            for (int i = 0; i <= 10; i++) { c[i] = c10[i] / mx; rec[(size_t)i * S] = c[i]; }
            flag = !(fabs(c[10]) < 1e-15) ? 1 : 2;
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int k = 0; k < 4; k++) { rec[(size_t)(HR_BX + 4 * i + k) * S] = Bx[i][k]; rec[(size_t)(HR_BY + 4 * i + k) * S] = By[i][k]; }
#pragma unroll
            for (int k = 0; k < 5; k++) rec[(size_t)(HR_B1 + 5 * i + k) * S] = B1[i][k];
        }
        reinterpret_cast<int32_t*>(hyp + (size_t)HR_DOUBLES * S)[slot] = flag;
    }
    }
    HYP_SYNC();                                                   // k_ransac_hyp_list re-enters with the same LDS
}

// first chunk: grid (ceil(first / 16), npairs)
__global__ __launch_bounds__(64) void k_ransac_hyp(PoseParams P, int h0, int h_end, int npairs,
                                                   const double* __restrict__ n1, const double* __restrict__ n2,
                                                   const int32_t* __restrict__ samples, const int32_t* __restrict__ rstate,
                                                   double* __restrict__ hyp, size_t S) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ldsM = reinterpret_cast<double*>(smem);
    ransac_hyp_quad(P, blockIdx.y, h0 + blockIdx.x * QH, h_end, npairs, n1, n2, samples, rstate, hyp, S, ldsM, ldsM + HYP_STAGE * QH);
}

// Later chunks (h >= first) are only needed for the pairs whose adaptive bound is still above `first` after the first scan:
// k_ransac_scan appends those pairs to a work list and a fixed grid walks (pair, 16-hypothesis chunk) items, so the common
// case "nothing left to do" costs a handful of workgroups that exit immediately.  chunks = 64-hypothesis chunks per pair
// (the unit of the roots / score kernels' sub-items): four 16-hypothesis items each.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_ransac_hyp_list(PoseParams P, int h0, int h_end, int npairs,
                                                        const double* __restrict__ n1, const double* __restrict__ n2,
                                                        const int32_t* __restrict__ samples, const int32_t* __restrict__ rstate,
                                                        double* __restrict__ hyp, size_t S,
                                                        const int32_t* __restrict__ worklist, int chunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* ldsM = reinterpret_cast<double*>(smem);
    const int per_pair = chunks * (64 / QH);
    const int total = worklist[0] * per_pair;
    for (int wi = blockIdx.x; wi < total; wi += gridDim.x) {
        const int pair = __builtin_amdgcn_readfirstlane(worklist[1 + wi / per_pair]);      // wave-uniform: keep them in SGPRs
        const int hb = __builtin_amdgcn_readfirstlane(h0 + (wi % per_pair) * QH);
        ransac_hyp_quad(P, pair, hb, h_end, npairs, n1, n2, samples, rstate, hyp, S, ldsM, ldsM + HYP_STAGE * QH);
    }
}

// ---- sub-items: 16 consecutive hypotheses of one pair.  First chunk (worklist == nullptr): sub = pair, h in [0, 16).
// Later chunks: sub -> work-list entry (sub / (4*chunks)), 64-hypothesis chunk, quarter.
DEV bool sub_item(const int32_t* worklist, int chunks, int h0, int npairs, int sub, int& pair, int& hbase) {
    if (!worklist) { pair = sub; hbase = h0; return sub < npairs; }
    const int item = sub >> 2, q = sub & 3;
    if (item >= worklist[0] * chunks) return false;
    pair = worklist[1 + item / chunks];
    hbase = h0 + (item % chunks) * 64 + q * 16;
    return true;
}

// per lane: bit of `mask` set ? a : b.  Written out as two VOP3-encoded v_cndmask_b32 with the mask in an SGPR pair: the compiler
// picks the VOP2 encoding (selector in vcc) for a run of selects on one condition, and three or more of those back to back issue
// at 23 cycles each on gfx950 instead of 4 (profiles/r02_valu_cnd_mi355x.txt) -- in the bisection loop that was more than the
// Horner chain itself.
DEV double sel_f64(unsigned long long mask, double a, double b) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    unsigned lo, hi;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(lo) : "v"((unsigned)ub), "v"((unsigned)ua), "s"(mask));
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(hi) : "v"((unsigned)(ub >> 32)), "v"((unsigned)(ua >> 32)), "s"(mask));
    return __builtin_bit_cast(double, (unsigned long long)lo | ((unsigned long long)hi << 32));
}

// One bisection step on (lo, hi) given the mid point m and fm = f(m).  The sequential algorithm leaves its loop when the mid
// point no longer lies strictly inside (m == lo or m == hi: the interval has collapsed to neighbouring doubles); here the step is
// simply taken, because it is then a no-op: lo only ever moves to points where (f < 0) == neg and hi to points where it differs
// (true of the start values by the sign-change test), f(m) is the same Horner chain that classified that end point, so m == lo
// re-assigns lo and m == hi re-assigns hi.  That removes two f64 compares and the guard logic from every step; whether ALL active
// lanes of the wave have collapsed is looked at every eighth step only (extra steps change nothing).  Lanes without an interval
// run on garbage that nobody reads.
#define BISECT_STEP(act_)                                                               \
        if ((it & 7) == 7 && !__any((act_) && (m > lo) && (m < hi))) break;             \
        { const unsigned long long left_ = __builtin_amdgcn_ballot_w64(fm < 0) ^ negmask; lo = sel_f64(left_, m, lo); hi = sel_f64(left_, hi, m); }

// one bisection level of the derivative-interlacing root finder (oracle/pose.cpp real_roots), ONE candidate
// interval per lane: lane j of a 16-lane group owns the interval (prev[j-1], prev[j]) of the degree-D derivative.
// The arithmetic per interval is exactly the sequential algorithm's; only the mapping to lanes differs.
template <int D>
DEV void roots_level(const double (&c)[11], double B, int j, int gshift, double* prev, int& nprev) {
    constexpr int K = 10 - D;
    double q[D + 1];
#pragma unroll
    for (int i = 0; i <= D; i++) {
        double f = 1.0;
#pragma unroll
        for (int jj = 0; jj < K; jj++) f *= (double)(i + K - jj);
        q[i] = c[i + K] * f;
    }
    const bool mine = j < D && j <= nprev;
    double lo = (j == 0) ? -B : prev[j > 0 ? (j <= 10 ? j - 1 : 9) : 0];
    double hi = (j == nprev) ? B : prev[j < 10 ? j : 9];
    double flo = q[D], fhi = q[D];
#pragma unroll
    for (int i = D - 1; i >= 0; i--) { flo = flo * lo + q[i]; fhi = fhi * hi + q[i]; }
    const bool neg = flo < 0;
    const unsigned long long negmask = ~__builtin_amdgcn_ballot_w64(neg);      // left = (fm < 0) == neg
    const bool act = mine && ((flo < 0) != (fhi < 0));
    constexpr int NIT = (D == 10) ? BISECT_FINAL : BISECT_INNER;
    for (int it = 0; it < NIT; it++) {
        const double m = 0.5 * (lo + hi);
        double fm = q[D];
#pragma unroll
        for (int i = D - 1; i >= 0; i--) fm = fm * m + q[i];
        BISECT_STEP(act)
    }
    const double r = 0.5 * (lo + hi);
    const uint32_t mask = (uint32_t)(__builtin_amdgcn_ballot_w64(act) >> gshift) & 0xFFFFu;
    HYP_SYNC();                                                    // every lane has read prev[]
    if (act) prev[__popc(mask & ((1u << j) - 1u))] = r;
    nprev = __popc(mask);
    HYP_SYNC();
}

// Polynomials whose degree-10 coefficient vanished (record flag 2; S-752's planar, purely translating scenes produce one in every
// tenth 64-hypothesis item) take the SAME fixed-degree levels: oracle/pose.cpp real_roots() drops the leading coefficients below
// 1e-15 and works on degree deg = 10 - s.  With those coefficients set to exactly 0 the template's level D = d + s IS the reduced
// polynomial's level d -- the same derivative order 10 - D = deg - d, the same coefficients, a Horner chain that starts on zeros
// (0 * m + q[d] == q[d] bit for bit), 200 bisection steps on the last level and 40 on the others -- and the levels D <= s see a
// constant or the zero polynomial: no sign change, no roots, which is the state level d = 1 starts from.  Only the root bound B
// differs (coefficients relative to c[deg]).  Returns B; c is modified in place.
DEV double poly_prepare(double (&c)[11], int flag) {
    int deg = 10;
    if (flag == 2) {
#pragma unroll
        for (int i = 10; i >= 1; i--) if (deg == i && fabs(c[i]) < 1e-15) deg = i - 1;
    }
    double cd = c[10];
#pragma unroll
    for (int i = 9; i >= 0; i--) cd = (deg == i) ? c[i] : cd;
#pragma unroll
    for (int i = 1; i <= 10; i++) c[i] = (i > deg) ? 0.0 : c[i];
    double B = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) { const double t = fabs(c[i] / cd); B = (i < deg) ? fmax(B, t) : B; }
    return B + 1.0;
}

// real roots of every hypothesis polynomial: 16 lanes per hypothesis, 16 hypotheses (one sub-item) per block
__global__ __launch_bounds__(256) void k_hyp_roots(PoseParams P, int h0, int h_end, int npairs, const int32_t* __restrict__ rstate,
                                                   double* __restrict__ hyp, size_t S, const int32_t* __restrict__ worklist, int chunks) {
    __shared__ double sh_prev[16][10];
    const int g = threadIdx.x >> 4, j = threadIdx.x & 15, gshift = (threadIdx.x & 63) & ~15;
    for (int sub = blockIdx.x; ; sub += gridDim.x) {
        int pair, hbase;
        if (!sub_item(worklist, chunks, h0, npairs, sub, pair, hbase)) return;
        const int h = hbase + g;
        const bool active = h < rstate[(size_t)pair * RS] && h < h_end && h < max(P.max_iters, 1);
        const size_t slot = (size_t)pair * P.max_iters + (active ? h : min(hbase, max(P.max_iters, 1) - 1));   // inactive lanes never dereference it; kept in range anyway
        int32_t* flags = reinterpret_cast<int32_t*>(hyp + (size_t)HR_DOUBLES * S);
        int32_t* nrs = flags + S;
        const int flag = active ? flags[slot] : 0;
        double c[11];
#pragma unroll
        for (int i = 0; i <= 10; i++) c[i] = flag ? hyp[slot + (size_t)i * S] : (i == 10 ? 1.0 : 0.0);
        double* prev = sh_prev[g];
        int np = 0;
        if (__any(flag != 0)) {
            const double B = poly_prepare(c, flag);                // groups without a model: c = x^10, B = 1, results unused
            roots_level<1>(c, B, j, gshift, prev, np); roots_level<2>(c, B, j, gshift, prev, np);
            roots_level<3>(c, B, j, gshift, prev, np); roots_level<4>(c, B, j, gshift, prev, np);
            roots_level<5>(c, B, j, gshift, prev, np); roots_level<6>(c, B, j, gshift, prev, np);
            roots_level<7>(c, B, j, gshift, prev, np); roots_level<8>(c, B, j, gshift, prev, np);
            roots_level<9>(c, B, j, gshift, prev, np); roots_level<10>(c, B, j, gshift, prev, np);
        }
        if (flag) {
            if (j < np) hyp[slot + (size_t)(HR_ROOTS + j) * S] = prev[j];
            if (j == 0) nrs[slot] = np;
        } else if (active && j == 0) nrs[slot] = 0;
        HYP_SYNC();
        if (!worklist) return;
    }
}

// ---- the same root finder with ONE HYPOTHESIS PER LANE for the owner's part and the sign-change intervals of a level PACKED over
// the lanes, for batches (hundreds of hypotheses per pair, or 16 of each of four pairs in the first chunk: throughput matters, not
// latency).  With 16 lanes per polynomial every level costs its 40 bisection steps whatever the number of sign-change intervals,
// and most of the 16 lanes idle.  A wave that walks "interval j of my polynomial" for 64 polynomials runs max-over-lanes(number of
// intervals) rounds per level: measured on the fixed-1000 leg 88 % of the lane-rounds hold an interval at level 2, 52 % at levels
// 7-9, 66 % at the last one (62 % weighted by the work per round).  Here the owners list their intervals (owner lane, interval,
// root slot, sign at the left end, "right end is +B"), any lane bisects any interval with the owner's coefficients (LDS, read once
// per interval) and writes the root where the owner expects it (the other one of two root buffers: every interval of a level reads
// its end points from the previous level's).  A lane takes up to FOUR intervals per round (independent Horner chains issued
// side by side; 0.2 ms faster than two on the fixed-1000 leg).  The arithmetic per interval is the sequential algorithm's
// (oracle/pose.cpp real_roots): same coefficients, same end points, same Horner chain, same number of steps.
struct RootsLds {
    double cb[12][64];                                             // c[0..10] and the root bound B of every hypothesis of the wave
    double prev[2][10][64];                                        // roots of the previous / this level ([root][owner lane])
    uint16_t items[640];                                           // owner | interval << 6 | root slot << 10 | left sign << 14 | right end is +B << 15
};
// exact IEEE double multiply / add as instructions the compiler may not reorder: the C chains of a lane are issued as C multiplies, then
// C adds, so that a dependent pair is C - 1 instructions apart (the compiler's own order, mul a, add a, mul b, add b, stalls on
// every second instruction; in-order issue cannot hide that with two waves per SIMD)
DEV double mul_f64_ordered(double a, double b) { double r; asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV double add_f64_ordered(double a, double b) { double r; asm volatile("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int D, int C>
DEV void roots_bisect_packed(RootsLds& L, int k0, int N, int lane) {
    constexpr int K = 10 - D, NIT = (D == 10) ? BISECT_FINAL : BISECT_INNER;
    const double (*prev)[64] = L.prev[(D - 1) & 1];
    double (*next)[64] = L.prev[D & 1];
    bool v[C]; uint32_t d[C]; int h[C];
    double q[C][D + 1], lo[C], hi[C];
    unsigned long long nm[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int k = k0 + 64 * c + lane;
        v[c] = k < N;
        d[c] = L.items[v[c] ? k : k0];
        h[c] = d[c] & 63;
        const int j = (d[c] >> 6) & 15;
        const double B = L.cb[11][h[c]];
        lo[c] = j == 0 ? -B : prev[j > 0 ? j - 1 : 0][h[c]];
        hi[c] = (d[c] >> 15) ? B : prev[min(j, 9)][h[c]];
        nm[c] = ~__builtin_amdgcn_ballot_w64((d[c] >> 14) & 1u);    // left = (fm < 0) == neg
#pragma unroll
        for (int i = 0; i <= D; i++) {
            double f = 1.0;
#pragma unroll
            for (int jj = 0; jj < K; jj++) f *= (double)(i + K - jj);
            q[c][i] = L.cb[i + K][h[c]] * f;
        }
    }
    for (int it = 0; it < NIT; it++) {
        double m[C], f[C];
#pragma unroll
        for (int c = 0; c < C; c++) { m[c] = 0.5 * (lo[c] + hi[c]); f[c] = q[c][D]; }
#pragma unroll
        for (int i = D - 1; i >= 0; i--) {
#pragma unroll
            for (int c = 0; c < C; c++) f[c] = mul_f64_ordered(f[c], m[c]);
#pragma unroll
            for (int c = 0; c < C; c++) f[c] = add_f64_ordered(f[c], q[c][i]);
        }
        // (see BISECT_STEP: a collapsed interval's step is a no-op, looked at every eighth step)
        if ((it & 7) == 7) {
            bool open = false;
#pragma unroll
            for (int c = 0; c < C; c++) open |= v[c] && m[c] > lo[c] && m[c] < hi[c];
            if (!__any(open)) break;
        }
#pragma unroll
        for (int c = 0; c < C; c++) {
            const unsigned long long l_ = __builtin_amdgcn_ballot_w64(f[c] < 0) ^ nm[c];
            lo[c] = sel_f64(l_, m[c], lo[c]); hi[c] = sel_f64(l_, hi[c], m[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) if (v[c]) next[(d[c] >> 10) & 15][h[c]] = 0.5 * (lo[c] + hi[c]);
}
template <int D>
DEV void roots_level_packed(const double (&c)[11], double B, RootsLds& L, int lane, bool flagged, int& nprev) {
    constexpr int K = 10 - D;
    const double (*prev)[64] = L.prev[(D - 1) & 1];
    // the owner's part: signs at the interval end points -B, root 0 .. root nprev-1 of the previous level, +B (as roots_level_lane)
    uint32_t sneg = 0;
    {
        double q[D + 1];
#pragma unroll
        for (int i = 0; i <= D; i++) {
            double f = 1.0;
#pragma unroll
            for (int jj = 0; jj < K; jj++) f *= (double)(i + K - jj);
            q[i] = c[i + K] * f;
        }
#pragma unroll
        for (int j = 0; j <= D; j++) {
            const double x = j == 0 ? -B : ((D > 1 && j <= D - 1 && j <= nprev) ? prev[j - 1][lane] : B);
            double fx = q[D];
#pragma unroll
            for (int i = D - 1; i >= 0; i--) fx = fx * x + q[i];
            sneg |= (fx < 0 ? 1u : 0u) << j;
        }
    }
    const int last = min(nprev, D - 1);                            // index of the interval that ends at +B
    uint32_t todo = flagged ? (sneg ^ (sneg >> 1)) & ((2u << last) - 1u) : 0u;
    const int n = __popc(todo);
    int incl = n;                                                  // inclusive prefix sum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d); if (lane >= d) incl += v; }
    const int N = __builtin_amdgcn_readlane(incl, 63);
    {
        int slot = incl - n, i = 0;
        while (todo) {
            const int j = __builtin_ctz(todo);
            todo &= todo - 1u;
            L.items[slot + i] = (uint16_t)(lane | (j << 6) | (i << 10) | (((sneg >> j) & 1u) << 14) | ((j >= last ? 1u : 0u) << 15));
            i++;
        }
    }
    HYP_SYNC();
    for (int k0 = 0; k0 < N; ) {                                    // up to four intervals per lane and round
        const int left = N - k0;
        if (left > 192) { roots_bisect_packed<D, 4>(L, k0, N, lane); k0 += 256; }
        else if (left > 128) { roots_bisect_packed<D, 3>(L, k0, N, lane); k0 += 192; }
        else if (left > 64) { roots_bisect_packed<D, 2>(L, k0, N, lane); k0 += 128; }
        else { roots_bisect_packed<D, 1>(L, k0, N, lane); k0 += 64; }
    }
    HYP_SYNC();
    nprev = n;
}

__global__ __launch_bounds__(64) void k_hyp_roots_packed(PoseParams P, int h0, int h_end, int npairs, const int32_t* __restrict__ rstate,
                                                         double* __restrict__ hyp, size_t S, const int32_t* __restrict__ worklist, int chunks) {
    __shared__ RootsLds L;
    const int lane = threadIdx.x;
    // items: 64 consecutive hypotheses of one work-list pair, or (first chunk, worklist == nullptr) 16 of each of four pairs
    const int total = worklist ? worklist[0] * chunks : (npairs + 3) / 4;
    int32_t* flags = reinterpret_cast<int32_t*>(hyp + (size_t)HR_DOUBLES * S);
    int32_t* nrs = flags + S;
    // Work-list items differ in their numbers of real roots: a workgroup CLAIMS its next item (counter behind the list, zeroed by
    // the scan that built the list) instead of walking a stride -- a grid of what fits on the chip stays busy to the end, and an
    // almost empty list (the adaptive headline) costs that many empty workgroups, not one per possible item.
    // (The first item is the workgroup's own index: with an almost empty list 2048 atomics on one address would cost 35 us.)
    int32_t* claim = worklist ? const_cast<int32_t*>(worklist) + 1 + npairs : nullptr;
    auto next_item = [&](int prev) -> int {
        if (!claim || prev < 0) return prev < 0 ? (int)blockIdx.x : prev + (int)gridDim.x;
        int it = 0;
        if (lane == 0) it = atomicAdd(claim, 1);
        return (int)gridDim.x + __builtin_amdgcn_readfirstlane(it);
    };
    for (int item = next_item(-1); item < total; item = next_item(item)) {
        const int pair_raw = worklist ? worklist[1 + item / chunks] : item * 4 + (lane >> 4);
        const int pair = min(pair_raw, npairs - 1);
        const int h = worklist ? h0 + (item % chunks) * 64 + lane : h0 + (lane & 15);
        const bool active = pair_raw < npairs && h < rstate[(size_t)pair * RS] && h < h_end && h < max(P.max_iters, 1);
        const size_t slot = (size_t)pair * P.max_iters + (active ? h : 0);       // inactive lanes never dereference it; kept in range anyway
        const int flag = active ? flags[slot] : 0;
        double c[11];
#pragma unroll
        for (int i = 0; i <= 10; i++) c[i] = flag ? hyp[slot + (size_t)i * S] : (i == 10 ? 1.0 : 0.0);
        int np = 0;
        if (__any(flag != 0)) {
            const double B = poly_prepare(c, flag);
            HYP_SYNC();                                            // the previous item's roots have been read
#pragma unroll
            for (int i = 0; i <= 10; i++) L.cb[i][lane] = c[i];
            L.cb[11][lane] = B;
            const bool fl = flag != 0;
            roots_level_packed<1>(c, B, L, lane, fl, np); roots_level_packed<2>(c, B, L, lane, fl, np);
            roots_level_packed<3>(c, B, L, lane, fl, np); roots_level_packed<4>(c, B, L, lane, fl, np);
            roots_level_packed<5>(c, B, L, lane, fl, np); roots_level_packed<6>(c, B, L, lane, fl, np);
            roots_level_packed<7>(c, B, L, lane, fl, np); roots_level_packed<8>(c, B, L, lane, fl, np);
            roots_level_packed<9>(c, B, L, lane, fl, np); roots_level_packed<10>(c, B, L, lane, fl, np);
        }
        if (flag) {
            for (int j = 0; j < np; j++) hyp[slot + (size_t)(HR_ROOTS + j) * S] = L.prev[0][j][lane];      // level 10 writes buffer 10 & 1 = 0
            nrs[slot] = np;
        } else if (active) nrs[slot] = 0;
    }
}

// Candidate essential matrices of every hypothesis: ONE HYPOTHESIS PER LANE, the lane walks over its real roots in ascending order
// and back-substitutes each (oracle/pose.cpp five_point(), the part after the roots).  A flat grid without LDS or barriers: the record
// fields of 64 consecutive hypotheses are 512 contiguous bytes per load.  (As the prologue of k_hyp_score -- thread = (hypothesis,
// root), 16 hypotheses per workgroup, the whole chain in front of three barriers -- this part took 0.7 of that kernel's 0.97 ms.)
// Valid models go to models[slot][m] in root order; the record's root-count word is overwritten by the number of valid models.
__global__ __launch_bounds__(256) void k_hyp_models(PoseParams P, int h0, int h_end, int npairs, const int32_t* __restrict__ rstate,
                                                    double* __restrict__ hyp, size_t S, double* __restrict__ models,
                                                    const int32_t* __restrict__ worklist, int chunks) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int32_t* nrs = reinterpret_cast<int32_t*>(hyp + (size_t)HR_DOUBLES * S) + S;
    const int total = worklist ? worklist[0] * chunks : (npairs + 3) / 4;      // items: 64 hypotheses of one pair / 16 of each of four pairs
    for (int item = blockIdx.x * 4 + wv; item < total; item += gridDim.x * 4) {
        int pair, h;
        if (worklist) { pair = worklist[1 + item / chunks]; h = h0 + (item % chunks) * 64 + lane; }
        else { pair = item * 4 + (lane >> 4); h = h0 + (lane & 15); }
        const bool active = pair < npairs && h < rstate[(size_t)min(pair, npairs - 1) * RS] && h < h_end && h < max(P.max_iters, 1);
        if (!active) continue;
        const size_t slot = (size_t)pair * P.max_iters + h;
        const int nr = nrs[slot];
        int m = 0;
        if (nr > 0) {
            const double* rec = hyp + slot;
            double cf[39], lb[36];
#pragma unroll
            for (int i = 0; i < 39; i++) cf[i] = rec[(size_t)(HR_BX + i) * S];          // Bx[3][4], By[3][4], B1[3][5]
#pragma unroll
            for (int i = 0; i < 36; i++) lb[i] = rec[(size_t)(HR_LB + i) * S];
            double* mo = models + slot * 90;
            for (int ri = 0; ri < nr; ri++) {
                const double z = rec[(size_t)(HR_ROOTS + ri) * S];
                double Bz[3][3];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    double v = cf[4 * i + 3];
#pragma unroll
                    for (int k = 2; k >= 0; k--) v = v * z + cf[4 * i + k];
                    Bz[i][0] = v;
                    v = cf[12 + 4 * i + 3];
#pragma unroll
                    for (int k = 2; k >= 0; k--) v = v * z + cf[12 + 4 * i + k];
                    Bz[i][1] = v;
                    v = cf[24 + 5 * i + 4];
#pragma unroll
                    for (int k = 3; k >= 0; k--) v = v * z + cf[24 + 5 * i + k];
                    Bz[i][2] = v;
                }
                double c01[3], c02[3], c12[3];
                cross3(Bz[0], Bz[1], c01); cross3(Bz[0], Bz[2], c02); cross3(Bz[1], Bz[2], c12);
                const double n01 = dot3(c01, c01), n02 = dot3(c02, c02), n12 = dot3(c12, c12);
                double nv0 = c01[0], nv1 = c01[1], nv2 = c01[2], nn = n01;
                if (n02 > nn) { nv0 = c02[0]; nv1 = c02[1]; nv2 = c02[2]; nn = n02; }
                if (n12 > nn) { nv0 = c12[0]; nv1 = c12[1]; nv2 = c12[2]; nn = n12; }
                if (!(nn > 0)) continue;
                const double inv = 1.0 / sqrt(nn);
                const double w = nv2 * inv;
                if (fabs(w) < 1e-10) continue;
                const double x = (nv0 * inv) / w, y = (nv1 * inv) / w;
                double E[9], fn = 0;
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    E[i] = ((x * lb[i] + y * lb[9 + i]) + z * lb[18 + i]) + lb[27 + i];
                    fn += E[i] * E[i];
                }
                fn = sqrt(fn);
                if (!(fn > 0)) continue;
#pragma unroll
                for (int i = 0; i < 9; i++) mo[9 * m + i] = E[i] / fn;
                m++;
            }
        }
        nrs[slot] = m;                                             // from here on: the number of candidate models
    }
}

// models + scores of one sub-item (16 hypotheses of one pair): thread (hyp, root) back-substitutes its root, the
// 256 threads then score every model of the sub-item against the pair's points (wave = model, lanes = points),
// and thread hyp picks the first model with the largest count.  hbest[pair][h] = (best count << 4) | model, -1 = none.
#define SC_CH 256                                                  // points staged in LDS per pass (512: 29 KB of LDS per workgroup and 1.30 instead of 0.98 ms per 1.02 M hypotheses)
// The Sampson test `(float)(num / den) <= thr^2` of cv's five-point estimator, decided in SINGLE precision wherever single precision
// can decide it, in double (the oracle's operation sequence, bit for bit) everywhere else.  FP64 issues at half the FP32 rate and
// the double form has no fused multiply-adds (the oracle is compiled without contraction): 34 FP64 operations against 20 FP32 ones + two comparisons.
//   s = x2^T E x1, den = (E x1)_0^2 + (E x1)_1^2 + (E^T x2)_0^2 + (E^T x2)_1^2, inlier <=> s^2 / den <= tmid (the double half way between
//   thr^2 and the next float).  With u = 2^-24, ||E||_F = 1 (k_hyp_models normalises), n1 = |(x1, y1, 1)|, n2 = |(x2, y2, 1)|, every float
//   input within u of its double, every fmaf rounding at most u times the sum of the magnitudes of the terms it has accumulated:
//     (E x1)_i from two fmaf: error <= 4 u A_i, A_i = |e_i0 x1| + |e_i1 y1| + |e_i2| <= |row i| n1; (E^T x2)_j likewise <= 4 u B_j, B_j <= |column j| n2
//     s: 4 u sum_i |x2_i| A_i + u (inputs x2, y2) + 2 u (its two fmaf), each times <= n1 n2 (Cauchy-Schwarz, ||E||_F = 1): < 7 u n1 n2   -> es = 8 u n1 n2
//     den: sum of 2 |v| 4 u A <= 8 u (n1^2 + n2^2) over its four squares + four roundings of partial sums <= n1^2 + n2^2: < 12 u (n1^2 + n2^2) -> ed = 16 u (n1^2 + n2^2)
//   (tests/test_independent_numpy.py: the largest errors seen on two million random pairs are 2.4 u n1 n2 and 3.3 u (n1^2 + n2^2))
//   With S, D the exact values: | (s32^2 - tmid den32) - (S^2 - tmid D) | <= es (2 |s32| + es) + tmid ed, and the three roundings of
//   r = fma(-tm, den32, s32 * s32) (tm = tmid rounded to float) add less than 2^-22 (s32^2 + tmid den32).  So with
//     band = kd den32 + 2 es' |s32| + c,   kd = 2^-14 tmid,  es' = es (1 + 2^-10),  c = (es^2 + tmid ed)(1 + 2^-10)
//   (two fmaf; the 2^-10 inflations absorb the roundings of band itself, kd den32 those of r and the 10^-15 relative error of the
//   double sequence it stands in for):   r <= -band  =>  certainly an inlier (and den32 > ed, so D > 0);   r >= band  =>  certainly
//   not.  Anything else -- a band of about +-0.02 % around the threshold residual at the image centre (+-0.1 % at the image corners
//   of config 3, n = 4), a NaN, a vanishing den -- takes the double path.  20 full-rate operations and two comparisons per (model, point).
//   The decisions, hence masks, counts and
//   iteration numbers, are the oracle's for every input (tests/test_pose_gpu.py, test_configs_gpu.py compare them exactly).
struct ScorePt { float x1, y1, x2, y2, es2, c; };
DEV int sampson_in_f64(const double* __restrict__ Em, double x1, double y1, double x2, double y2, double kLo, double kHi, float thr2) {
    const double Ex0 = (Em[0] * x1 + Em[1] * y1) + Em[2];
    const double Ex1 = (Em[3] * x1 + Em[4] * y1) + Em[5];
    const double Ex2 = (Em[6] * x1 + Em[7] * y1) + Em[8];
    const double Et0 = (Em[0] * x2 + Em[3] * y2) + Em[6];
    const double Et1 = (Em[1] * x2 + Em[4] * y2) + Em[7];
    const double x2tEx1 = (x2 * Ex0 + y2 * Ex1) + Ex2;
    const double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    const double num = x2tEx1 * x2tEx1, den = ((a + b) + c) + d;
    // division-free classification of (float)(num/den) <= thr2: num <= kLo*den is certainly an inlier, num >= kHi*den certainly an
    // outlier (margins 2^-40 >> the 2^-53 rounding of the products), anything in between takes the exact division
    if (den > 0 && num <= kLo * den) return 1;
    if (num >= kHi * den) return 0;
    return (float)(num / den) <= thr2 ? 1 : 0;
}
// single precision: in = certainly an inlier, out = certainly not, neither = undecided (a NaN fails both comparisons)
DEV void sampson_f32(const float (&e)[9], const ScorePt& p, float tm, float kd, bool& in, bool& out) {
    const float ex0 = __fmaf_rn(e[0], p.x1, __fmaf_rn(e[1], p.y1, e[2]));
    const float ex1 = __fmaf_rn(e[3], p.x1, __fmaf_rn(e[4], p.y1, e[5]));
    const float ex2 = __fmaf_rn(e[6], p.x1, __fmaf_rn(e[7], p.y1, e[8]));
    const float et0 = __fmaf_rn(e[0], p.x2, __fmaf_rn(e[3], p.y2, e[6]));
    const float et1 = __fmaf_rn(e[1], p.x2, __fmaf_rn(e[4], p.y2, e[7]));
    const float s = __fmaf_rn(p.x2, ex0, __fmaf_rn(p.y2, ex1, ex2));
    const float den = __fmaf_rn(ex0, ex0, __fmaf_rn(ex1, ex1, __fmaf_rn(et0, et0, et1 * et1)));
    const float r = __fmaf_rn(-tm, den, s * s);
    const float band = __fmaf_rn(kd, den, __fmaf_rn(p.es2, fabsf(s), p.c));
    in = r <= -band; out = r >= band;
}
// returns 1 / 0, or -1 = undecided in single precision
DEV int sampson_in_f32(const float (&e)[9], const ScorePt& p, float tm, float kd) {
    bool in, out; sampson_f32(e, p, tm, kd, in, out);
    return in ? 1 : (out ? 0 : -1);
}
// rstate is read (words 0, 6) AND written (word 8, the models-scored counter) here: a plain pointer, no const / __restrict__ promise
__global__ __launch_bounds__(256) void k_hyp_score(PoseParams P, int h0, int h_end, int npairs, int32_t* rstate,
                                                   const double* __restrict__ n1, const double* __restrict__ n2,
                                                   const double* __restrict__ hyp, size_t S, const double* __restrict__ models,
                                                   int32_t* __restrict__ hbest, const int32_t* __restrict__ worklist, int chunks) {
    __shared__ __attribute__((aligned(16))) float sE[160][12];     // the sub-item's models, single precision, 48-byte rows (the double path re-reads `models`)
    __shared__ ScorePt sP[SC_CH];
    // (model, point) decisions that single precision left open, settled afterwards by ALL threads at once: inside the model loop a
    // single undecided lane would hold its whole wave for a double-precision evaluation from global memory
    constexpr int AMB_CAP = 4096;
    static_assert(VIS_RANSAC_MAX_M <= 65536, "the undecided list packs the point index into 16 bits");
    __shared__ uint32_t sAmb[AMB_CAP];
    __shared__ int32_t sNamb;
    __shared__ int32_t sBase[16], sCnt[16], sTag[160], sGood[160], sTotal;
    const int tid = threadIdx.x, lane = tid & 63;
    const float thr2 = (float)(P.thr * P.thr);
    const double tmid = 0.5 * ((double)thr2 + (double)__uint_as_float(__float_as_uint(thr2) + 1u));
    const double kLo = tmid * (1.0 - 0x1p-40), kHi = tmid * (1.0 + 0x1p-40);
    const float tm = (float)tmid, kd = __double2float_ru(tmid * 0x1p-14), thi16 = __double2float_ru(tmid * (1.0 + 0x1p-16));
    for (int sub = blockIdx.x; ; sub += gridDim.x) {
        int pair, hbase;
        if (!sub_item(worklist, chunks, h0, npairs, sub, pair, hbase)) return;
        const int niters = rstate[(size_t)pair * RS], M = rstate[(size_t)pair * RS + 6];
        const int32_t* nmod = reinterpret_cast<const int32_t*>(hyp + (size_t)HR_DOUBLES * S) + S;    // models per hypothesis (k_hyp_models)
        if (tid < 16) {
            const int h = hbase + tid;
            const bool active = h < niters && h < h_end && h < max(P.max_iters, 1);
            sCnt[tid] = active ? nmod[(size_t)pair * P.max_iters + h] : 0;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0; for (int k = 0; k < 16; k++) { sBase[k] = t; t += sCnt[k]; } sTotal = t;
#ifndef VIS_SCORE_COUNT_UNDECIDED                                  // diagnostic build: n_models reports the undecided (model, point) decisions instead
            if (t) atomicAdd(rstate + (size_t)pair * RS + 8, t);   // point evaluations = models x M
#endif
        }
        __syncthreads();
        const double* mbase = models + ((size_t)pair * P.max_iters + hbase) * 90;
        {
            const int hy = tid / 10, m = tid - hy * 10;            // thread = (hypothesis, model): the sub-item's model list in LDS
            if (tid < 160 && m < sCnt[hy]) {
                const int t = sBase[hy] + m;
                const double* mo = mbase + (size_t)hy * 90 + 9 * m;
#pragma unroll
                for (int k = 0; k < 9; k++) sE[t][k] = (float)mo[k];
                sTag[t] = (hy << 4) | m; sGood[t] = 0;
            }
        }
        const int T = sTotal;
        const double* pa = n1 + (size_t)pair * P.mcap * 2;
        const double* pb = n2 + (size_t)pair * P.mcap * 2;
        // one (model t, point c0 + i) decision: single precision first, the oracle's double sequence when that cannot tell
        auto decide = [&](const float (&e)[9], const ScorePt& pt, int t, int c0, int i) -> int {
            int in = sampson_in_f32(e, pt, tm, kd);
            if (in < 0) {
                const int tag = sTag[t];
                const double* mo = mbase + (size_t)(tag >> 4) * 90 + 9 * (tag & 15);
                in = sampson_in_f64(mo, pa[2 * (c0 + i)], pa[2 * (c0 + i) + 1], pb[2 * (c0 + i)], pb[2 * (c0 + i) + 1], kLo, kHi, thr2);
            }
            return in;
        };
        auto make_pt = [&](int gi) -> ScorePt {                     // point gi of the pair in single precision + its error radii (see above)
            const double x1 = pa[2 * gi], y1 = pa[2 * gi + 1], x2 = pb[2 * gi], y2 = pb[2 * gi + 1];
            ScorePt q; q.x1 = (float)x1; q.y1 = (float)y1; q.x2 = (float)x2; q.y2 = (float)y2;
            const float n1 = sqrtf(__fmaf_rn(q.x1, q.x1, __fmaf_rn(q.y1, q.y1, 1.f))) * (1.f + 0x1p-20f);
            const float n2 = sqrtf(__fmaf_rn(q.x2, q.x2, __fmaf_rn(q.y2, q.y2, 1.f))) * (1.f + 0x1p-20f);
            const float es = 8.f * 0x1p-24f * n1 * n2, ed = 16.f * 0x1p-24f * (n1 * n1 + n2 * n2);
            q.es2 = 2.f * es * (1.f + 0x1p-10f); q.c = (es * es + thi16 * ed) * (1.f + 0x1p-10f);
            return q;
        };
        if (M <= 256) {
          for (int c0 = 0; c0 < (T > 0 ? M : 0); c0 += SC_CH) {
            const int mc = min(M - c0, SC_CH);
            __syncthreads();
            for (int i = tid; i < mc; i += 256) sP[i] = make_pt(c0 + i);
            __syncthreads();
            {
                // Few correspondences (the reference pipeline: <= root^2 = 49 grid matches): LANE = (model, slice of the points).  A
                // wave-per-model pass would leave a third of the lanes idle at M = 43 and pay a wave reduction per model; here the
                // only reduction is one LDS add per lane.  The points are cut into K slices, K chosen per sub-item so that the
                // T x K (model, slice) items fill the 256 lanes in the fewest passes: at the typical T = 67 models of 16 hypotheses
                // K = 3 takes one pass of 15 points where a fixed K = 4 (64 models per pass) took two passes of 11.
                int K = 1, cost = 0x7FFFFFFF;
                for (int k = 1; k <= 8; k++) {
                    const int per = 256 / k, c = ((T + per - 1) / per) * ((mc + k - 1) / k);
                    if (c < cost) { cost = c; K = k; }
                }
                const int per = 256 / K;
                const int ks = tid / per, tl = tid - ks * per;                        // slice, model within the pass
                const int q0 = (mc * ks) / K, q1 = ks < K ? (mc * (ks + 1)) / K : q0;   // lanes beyond per * K idle
                for (int t0 = 0; t0 < T; t0 += per) {
                    const int t = t0 + tl;
                    const int tc = min(t, T - 1);
                    float Em[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) Em[k] = sE[tc][k];
                    int good = 0;
                    for (int i = q0; i < q1; i++) good += decide(Em, sP[i], tc, c0, i);
                    if (t < T && good) atomicAdd(&sGood[t], good);
                }
            }
          }
        } else if (T > 0) {
            // Thousands of correspondences (BASELINE config 3: RANSAC on ~3100 symmetric matches): LANE = up to four points held in
            // registers (straight from global memory: no staging, no barrier), every thread walks ALL models of the sub-item (three
            // broadcast LDS reads per model); a model's count over the wave is one ballot + population count per point instead of a
            // wave reduction per (model, chunk).  The ceil(M / 256) rows of 256 points are dealt evenly over ceil(rows / 4) rounds
            // (M = 3100: 13 rows as 4 + 3 + 3 + 3, not 4 + 4 + 4 + 1).  What single precision cannot decide goes on a list
            // (model << 16 | point) and counts as "out" in the loop; ALL threads settle the list in double precision afterwards.
            if (tid == 0) sNamb = 0;
            __syncthreads();
            auto round_of = [&](auto ppl_tag, int g0, int mc) {    // PPL rows of 256 points starting at point g0, mc points in all
                constexpr int PPL = decltype(ppl_tag)::value;
                ScorePt Q[PPL]; unsigned long long vm[PPL];       // vm: lanes of the wave that hold a point (scalar masks)
#pragma unroll
                for (int k = 0; k < PPL; k++) {
                    const bool vq = tid + 256 * k < mc;
                    Q[k] = make_pt(g0 + (vq ? tid + 256 * k : 0));
                    // a lane without a point: band = -inf makes it a certain inlier of every finite model; the counts and the undecided
                    // list take only the lanes of vm (a NON-finite model -- fn = inf passes k_hyp_models' `fn > 0` -- gives r = NaN in every
                    // lane: such lanes used to be subtracted as inliers they never were and listed as undecided points beyond M)
                    if (!vq) { Q[k].x1 = Q[k].y1 = Q[k].x2 = Q[k].y2 = 0.f; Q[k].es2 = 0.f; Q[k].c = -__builtin_inff(); }
                    vm[k] = __builtin_amdgcn_ballot_w64(vq);
                }
                for (int t = 0; t < T; t++) {
                    const float4 r0 = *reinterpret_cast<const float4*>(&sE[t][0]), r1 = *reinterpret_cast<const float4*>(&sE[t][4]);
                    const float Em[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, sE[t][8]};
                    unsigned long long bd[PPL], alld = ~0ull; int cnt = 0;
#pragma unroll
                    for (int k = 0; k < PPL; k++) {                // branch-free: 20 fma/mul/add + two compares per point
                        bool in, out; sampson_f32(Em, Q[k], tm, kd, in, out);
                        const unsigned long long bi = __builtin_amdgcn_ballot_w64(in), bo = __builtin_amdgcn_ballot_w64(out);   // one compare each
                        bd[k] = bi | bo | ~vm[k]; alld &= bd[k];
                        cnt += __popcll(bi & vm[k]);
                    }
                    if (__builtin_expect(alld != ~0ull, 0)) {      // wave-uniform: ONE list reservation for all undecided lanes of the wave
                        int tot = 0;
#pragma unroll
                        for (int k = 0; k < PPL; k++) tot += __popcll(~bd[k]);
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&sNamb, tot);
                        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                        for (int k = 0; k < PPL; k++) {
                            const unsigned long long u = ~bd[k];
                            const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(u >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)u, 0u));
                            if (((u >> lane) & 1) && slot < AMB_CAP) sAmb[slot] = ((uint32_t)t << 16) | (uint32_t)(g0 + tid + 256 * k);
                            base += __popcll(u);
                        }
                    }
                    if (lane == 0 && cnt) atomicAdd(&sGood[t], cnt);
                }
            };
            const int rows = (M + 255) >> 8, rounds = (rows + 3) >> 2, rbase = rows / rounds, rrem = rows - rbase * rounds;
            int g0 = 0;
            for (int r = 0; r < rounds; r++) {
                const int nr = rbase + (r < rrem ? 1 : 0), mc = min(M - g0, nr * 256);
                if (nr == 4) round_of(std::integral_constant<int, 4>{}, g0, mc);
                else if (nr == 3) round_of(std::integral_constant<int, 3>{}, g0, mc);
                else if (nr == 2) round_of(std::integral_constant<int, 2>{}, g0, mc);
                else round_of(std::integral_constant<int, 1>{}, g0, mc);
                g0 += nr * 256;
            }
            __syncthreads();
            const int na = sNamb;
#ifdef VIS_SCORE_COUNT_UNDECIDED
            if (tid == 0) atomicAdd(rstate + (size_t)pair * RS + 8, na);
#endif
            if (__builtin_expect(na <= AMB_CAP, 1)) {
                for (int a = tid; a < na; a += 256) {
                    const uint32_t code = sAmb[a];
                    const int t = (int)(code >> 16), gi = (int)(code & 0xFFFFu), tag = sTag[t];
                    if (gi >= M) continue;                         // (a lane without a point: only a non-finite model can leave it undecided)
                    if (sampson_in_f64(mbase + (size_t)(tag >> 4) * 90 + 9 * (tag & 15), pa[2 * gi], pa[2 * gi + 1], pb[2 * gi], pb[2 * gi + 1], kLo, kHi, thr2))
                        atomicAdd(&sGood[t], 1);
                }
            } else {
                // the list overflowed (not seen: it holds 2 % of the decisions of 67 models x 3100 points, about 0.1 % are
                // undecided): every count of the sub-item again, in double
                __syncthreads();
                for (int t = tid; t < T; t += 256) sGood[t] = 0;
                __syncthreads();
                for (int t = 0; t < T; t++) {
                    const int tag = sTag[t];
                    const double* mo = mbase + (size_t)(tag >> 4) * 90 + 9 * (tag & 15);
                    int good = 0;
                    for (int gi = tid; gi < M; gi += 256)
                        good += sampson_in_f64(mo, pa[2 * gi], pa[2 * gi + 1], pb[2 * gi], pb[2 * gi + 1], kLo, kHi, thr2);
                    if (good) atomicAdd(&sGood[t], good);
                }
            }
        }
        __syncthreads();
        if (tid < 16) {
            const int h = hbase + tid;
            const bool active = h < niters && h < h_end && h < max(P.max_iters, 1);
            int bestc = -1, bestm = 0;
            for (int m = 0; m < sCnt[tid]; m++) { const int gd = sGood[sBase[tid] + m]; if (gd > bestc) { bestc = gd; bestm = m; } }
            if (active) hbest[(size_t)pair * P.max_iters + h] = sCnt[tid] ? ((bestc << 4) | bestm) : -1;
        }
        __syncthreads();
        if (!worklist) return;
    }
}

DEV int update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = fmax(p, 0.); p = fmin(p, 1.);
    ep = fmax(ep, 0.); ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num); denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

// replay RANSACPointSetRegistrator::run's accept/update rule over hypotheses [rs[4], min(hi, niters)).
// One wave per pair: 64 per-iteration results are fetched in parallel, then walked in order (all lanes
// carry the same scalar state).  Within one iteration only its best count matters: counts are accepted
// in model order when strictly greater, so the survivor is the first model reaching the iteration's
// maximum, and the iteration bound only ever shrinks with larger counts.
__global__ __launch_bounds__(64) void k_ransac_scan(PoseParams P, int hi, const int32_t* __restrict__ hbest, int32_t* __restrict__ rstate,
                                                    int32_t* __restrict__ worklist) {
    const int pair = blockIdx.x, lane = threadIdx.x;
    int32_t* rs = rstate + (size_t)pair * RS;
    // k_hyp_roots_packed's item counter behind the list: zeroed by the scan that builds the list (it runs once, before the work-list
    // kernels) -- here, in front of every early return (pair 0 may well be a pair without a model to estimate)
    if (worklist && pair == 0 && lane == 0) worklist[1 + gridDim.x] = 0;
    if (rs[5] != 0) {
        if (rs[5] == 1 && lane == 0) {
            // M == 5: one hypothesis from all five points, its FIRST model, an all-ones mask -- or nothing at all when the solver
            // found no model (oracle/pose.cpp: `if (nm <= 0) return`; hbest = -1 once the hypothesis has been scored, hi > 0)
            const bool none = hi > 0 && hbest[(size_t)pair * P.max_iters] < 0;
            rs[2] = none ? -1 : 0; rs[3] = 0; rs[1] = none ? 0 : 5; rs[7] = none ? 0 : 1; rs[4] = 1;
            if (worklist && hi == 0) worklist[1 + atomicAdd(&worklist[0], 1)] = pair;     // no first chunk ran: its one hypothesis is still to be solved
        }
        return;
    }
    int niters = rs[0], maxGood = rs[1], iter = rs[4], bi = rs[2], bm = rs[3];
    const int M = rs[6];
    const int32_t* hb = hbest + (size_t)pair * P.max_iters;
    // 256 iterations per round (four results per lane, loaded together).  Only an ACCEPTED iteration changes the state (its count
    // exceeds everything before it; the adaptive bound can only shrink then), so inside a 64-iteration block the walk jumps from
    // one accepted iteration to the next by ballot instead of visiting all 64: same survivor, same bound, same iteration count.
    while (iter < niters && iter < hi) {
        const int base0 = iter;
        int v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { const int idx = base0 + 64 * j + lane; v[j] = (idx < hi && idx < P.max_iters) ? hb[idx] : -1; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int base = base0 + 64 * j;
            if (!(base < niters && base < hi)) break;
            int lim = min(niters, hi) - base;                      // iterations of this block still inside the bound (>= 1)
            const int good_l = v[j] < 0 ? -1 : (v[j] >> 4);
            int start = 0, last = -1;
            for (;;) {
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(lane >= start && lane < lim && good_l > max(maxGood, 4));
                if (!mask) break;
                const int k = __builtin_ctzll(mask);
                const int e = __builtin_amdgcn_readlane(v[j], k);
                bi = base + k; bm = e & 15; maxGood = e >> 4;
                if (P.adaptive) { niters = update_num_iters(P.prob, (double)(M - maxGood) / M, 5, niters); lim = min(niters, hi) - base; }
                last = k; start = k + 1;
            }
            iter = base + max(min(64, lim), last + 1);             // the sequential loop leaves one past the last iteration it ran
        }
    }
    if (lane == 0) {
        rs[0] = niters; rs[1] = maxGood; rs[2] = bi; rs[3] = bm; rs[4] = iter; rs[7] = iter;
        if (worklist && niters > hi && hi < P.max_iters) worklist[1 + atomicAdd(&worklist[0], 1)] = pair;   // needs more hypotheses
    }
}

__device__ void svd3_decompose(const double* E, double* U, double* Vt) {
    double A[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += E[3 * k + i] * E[3 * k + j];
        A[3 * i + j] = s;
    }
    jacobi_eig<3>(A, V);
    int ord[3] = {0, 1, 2};
    // sort by descending eigenvalue, ties keep index order
    for (int i = 0; i < 3; i++) for (int j = i + 1; j < 3; j++) {
        const bool sw = (A[4 * ord[j]] > A[4 * ord[i]]) || (A[4 * ord[j]] == A[4 * ord[i]] && ord[j] < ord[i]);
        if (sw) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    }
    double v0[3], v1[3], v2[3];
    for (int k = 0; k < 3; k++) { v0[k] = V[3 * k + ord[0]]; v1[k] = V[3 * k + ord[1]]; }
    cross3(v0, v1, v2);
    double u0[3], u1[3], u2[3];
    for (int r = 0; r < 3; r++) { u0[r] = dot3(E + 3 * r, v0); u1[r] = dot3(E + 3 * r, v1); }
    const double n0 = sqrt(dot3(u0, u0));
    for (int r = 0; r < 3; r++) u0[r] /= n0;
    const double pr = dot3(u0, u1);
    for (int r = 0; r < 3; r++) u1[r] -= pr * u0[r];
    const double nn1 = sqrt(dot3(u1, u1));
    for (int r = 0; r < 3; r++) u1[r] /= nn1;
    cross3(u0, u1, u2);
    for (int r = 0; r < 3; r++) { U[3 * r] = u0[r]; U[3 * r + 1] = u1[r]; U[3 * r + 2] = u2[r]; }
    for (int c = 0; c < 3; c++) { Vt[c] = v0[c]; Vt[3 + c] = v1[c]; Vt[6 + c] = v2[c]; }
}

DEV void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
        C[3 * i + j] = s;
    }
}

// DLT triangulation of one correspondence under P = [R | t] (4 x 4 Jacobi eigen-decomposition of A^T A, eigenvector of the smallest
// eigenvalue) + the cheirality test of recoverPose (depth positive and below 50 in both cameras); X_out = the homogeneous point
__device__ __forceinline__ bool cheirality(const double* R, const double* t, double x1, double y1, double x2, double y2, double* X_out, bool* zero_theta) {
    const double P[12] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2]};
    double A[16];
    A[0] = -1; A[1] = 0;  A[2] = x1; A[3] = 0;
    A[4] = 0;  A[5] = -1; A[6] = y1; A[7] = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { A[8 + c] = x2 * P[8 + c] - P[c]; A[12 + c] = y2 * P[8 + c] - P[4 + c]; }
    double AtA[16], V[16];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) s += A[4 * k + i] * A[4 * k + j];
            AtA[4 * i + j] = s;
        }
    jacobi_eig<4>(AtA, V, zero_theta);
    // eigenvector of the smallest eigenvalue (first one on ties), picked with selects: no run-time index into the register arrays
    int mn = 0; double dmin = AtA[0];
#pragma unroll
    for (int i = 1; i < 4; i++) if (AtA[5 * i] < dmin) { dmin = AtA[5 * i]; mn = i; }
    double X[4];
#pragma unroll
    for (int k = 0; k < 4; k++) X[k] = mn == 0 ? V[4 * k] : mn == 1 ? V[4 * k + 1] : mn == 2 ? V[4 * k + 2] : V[4 * k + 3];
    if (X_out) { X_out[0] = X[0]; X_out[1] = X[1]; X_out[2] = X[2]; X_out[3] = X[3]; }
    bool ok = (X[2] * X[3]) > 0;
    const double Xn[3] = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};
    ok = ok && (Xn[2] < 50.0);
    const double z2 = ((P[8] * Xn[0] + P[9] * Xn[1]) + P[10] * Xn[2]) + P[11];
    ok = ok && (z2 > 0) && (z2 < 50.0);
    return ok;
}

// The votes of one correspondence for [R | t] AND [R | -t] from ONE eigen-decomposition.  Negating t negates column 3 of the DLT
// matrix (its first two rows have a zero there), i.e. A' = A D with D = diag(1, 1, 1, -1), so A'^T A' = D (A^T A) D: every entry of
// the matrix, and of everything the Jacobi rotations compute from it, is the same number with the sign D gives it -- IEEE add,
// multiply, divide and sqrt commute with negation exactly, and the convergence tests look at squares and magnitudes only.  The
// eigenvector of the second problem is therefore X' = D X bit for bit, and its test runs on (X0, X1, X2, -X3) with -t.  The one
// operation that is not odd is `theta >= 0 ? 1 : -1` at theta == +-0 (two equal diagonal entries in a rotation with column 3);
// the decomposition reports that case and the second candidate is then decomposed on its own, as the oracle does.
__device__ __forceinline__ void cheirality_pair(const double* R, const double* t, double x1, double y1, double x2, double y2, bool& ok_pos, bool& ok_neg) {
    double tt[3] = {t[0], t[1], t[2]};
    ok_pos = ok_neg = false;
#pragma nounroll
    for (int pass = 0; pass < 2; pass++) {                         // ONE copy of the decomposition in the code: the second pass is the rare fallback
        double X[4]; bool zt = false;
        const bool ok = cheirality(R, tt, x1, y1, x2, y2, X, &zt);
        tt[0] = -tt[0]; tt[1] = -tt[1]; tt[2] = -tt[2];
        if (pass == 1) { ok_neg = ok; break; }
        ok_pos = ok;
        if (!__any(zt)) {                                          // (wave-uniform exit: lanes with the special case carry the others through the second pass)
            const double X3 = -X[3];
            bool o = (X[2] * X3) > 0;
            const double Xn[3] = {X[0] / X3, X[1] / X3, X[2] / X3};
            o = o && (Xn[2] < 50.0);
            const double z2 = ((R[6] * Xn[0] + R[7] * Xn[1]) + R[8] * Xn[2]) + tt[2];
            ok_neg = o && (z2 > 0) && (z2 < 50.0);
            break;
        }
    }
}

// recoverPose, part 1: the four (R, t) candidates of every pair's essential matrix (Jacobi SVD, decomposeEssentialMat), ONE PAIR
// PER LANE.  As thread 0 of k_pose_final's per-pair workgroup this serial chain kept four waves' registers resident for tens of
// microseconds each -- on the low-priority stream, but in the way of the detect kernels' workgroups (the headline moved by 2 % with
// this kernel's register count).  cand[pair]: R1 (9), R2 (9), t (3) doubles -- kept in the hypothesis buffer, which is free by now.
__global__ __launch_bounds__(64) void k_pose_svd(PoseParams P, const double* __restrict__ models, const int32_t* __restrict__ rstate,
                                                 const double* __restrict__ E_in, double* __restrict__ cand, int npairs) {
    const int pair = blockIdx.x * 64 + threadIdx.x;
    if (pair >= npairs) return;
    const int32_t* rs = rstate + (size_t)pair * RS;
    const bool have = E_in ? true : (rs[2] >= 0);
    if (!have) return;
    double E[9];
#pragma unroll
    for (int i = 0; i < 9; i++) E[i] = E_in ? E_in[(size_t)pair * 9 + i] : models[((size_t)pair * P.max_iters + rs[2]) * 90 + 9 * rs[3] + i];
    double U[9], Vt[9]; svd3_decompose(E, U, Vt);
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9], R1[9], R2[9];
    mat3_mul(U, W, T); mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T); mat3_mul(T, Vt, R2);
    double* c = cand + (size_t)pair * 21;
#pragma unroll
    for (int i = 0; i < 9; i++) { c[i] = R1[i]; c[9 + i] = R2[i]; }
    c[18] = U[2]; c[19] = U[5]; c[20] = U[8];
}

// winner's inlier mask + recoverPose.  do_pose = 0 -> only the mask / E (findEssentialMat).  grid (pairs, nsplit): the M Sampson
// tests and the 4 M triangulations (a 4 x 4 Jacobi eigen-decomposition each) of a pair are dealt over nsplit workgroups -- with
// thousands of correspondences per pair (BASELINE config 3: M ~ 3100, 32 pairs) one workgroup per pair left the chip idle for
// 1.6 ms.  Votes and the inlier count are integers summed with atomics (order independent); the workgroup that finishes last
// (ticket in rstate[14]) writes the pair's record.
__global__ __launch_bounds__(256) void k_pose_final(PoseParams P, const double* __restrict__ n1, const double* __restrict__ n2,
                                                    const double* __restrict__ models, int32_t* rstate,
                                                    const double* __restrict__ E_in, uint8_t* __restrict__ mask_out,
                                                    PoseOut* __restrict__ out, int do_pose, const double* __restrict__ cand) {
    __shared__ double sE[9], sR[2][9], sT[3];
    __shared__ int sgood[4], sinl;
    const int pair = blockIdx.x, tid = threadIdx.x, nsplit = gridDim.y, part = blockIdx.y;
    int32_t* rs = rstate + (size_t)pair * RS;
    const int M = rs[6];
    const double* a = n1 + (size_t)pair * P.mcap * 2;
    const double* b = n2 + (size_t)pair * P.mcap * 2;
    const bool have = E_in ? true : (rs[2] >= 0);
    if (tid < 9) sE[tid] = !have ? 0.0 : (E_in ? E_in[(size_t)pair * 9 + tid] : models[((size_t)pair * P.max_iters + rs[2]) * 90 + 9 * rs[3] + tid]);
    if (tid < 4) sgood[tid] = 0;
    if (tid == 0) sinl = 0;
    __syncthreads();
    const float t = (float)(P.thr * P.thr);
    if (!E_in) {
        for (int i = part * 256 + tid; i < M; i += 256 * nsplit) {
            int f = have ? sampson_inlier(sE, a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1], t) : 0;
            if (have && rs[5] == 1) f = 1;                         // M == 5: all-ones mask
            if (mask_out) mask_out[(size_t)pair * P.mcap + i] = (uint8_t)f;
            if (f) atomicAdd(&sinl, 1);
        }
    }
    if (tid < 21 && do_pose && have) {                              // the pair's candidates (k_pose_svd)
        const double v = cand[(size_t)pair * 21 + tid];
        if (tid < 18) sR[tid / 9][tid % 9] = v; else sT[tid - 18] = v;
    }
    __syncthreads();
    if (do_pose && have) {
        for (int w = part * 256 + tid; w < 2 * M; w += 256 * nsplit) {       // (rotation c, point i): votes for (R_c, t) and (R_c, -t)
            const int c = w / M, i = w - c * M;
            bool okp, okn;
            cheirality_pair(sR[c], sT, a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1], okp, okn);
            if (okp) atomicAdd(&sgood[c], 1);
            if (okn) atomicAdd(&sgood[c + 2], 1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        int last = 1;
        if (nsplit > 1) {
            for (int c = 0; c < 4; c++) if (sgood[c]) atomicAdd(rs + 9 + c, sgood[c]);
            __threadfence();                                       // the sums are visible before the ticket
            last = atomicAdd(rs + 14, 1) == nsplit - 1;
            if (last) { __threadfence(); for (int c = 0; c < 4; c++) sgood[c] = atomicAdd(rs + 9 + c, 0); }
        }
        if (last) {
            PoseOut o;
            for (int i = 0; i < 9; i++) { o.E[i] = sE[i]; o.R[i] = 0; }
            o.t[0] = o.t[1] = o.t[2] = 0;
            o.n_inliers = E_in ? 0 : (have ? (rs[5] == 1 ? 5 : rs[1]) : 0);
            o.iters_run = rs[7]; o.n_points = M; o.n_pose_good = 0; o.n_models = rs[8]; o.reserved_ = 0;
            if (do_pose && have) {
                const int* g = sgood;
                int sel;
                if (g[0] >= g[1] && g[0] >= g[2] && g[0] >= g[3]) sel = 0;
                else if (g[1] >= g[0] && g[1] >= g[2] && g[1] >= g[3]) sel = 1;
                else if (g[2] >= g[0] && g[2] >= g[1] && g[2] >= g[3]) sel = 2;
                else sel = 3;
                for (int i = 0; i < 9; i++) o.R[i] = sR[sel & 1][i];
                for (int i = 0; i < 3; i++) o.t[i] = sel < 2 ? sT[i] : -sT[i];
                o.n_pose_good = g[sel];
            }
            out[pair] = o;
        }
    }
}

// ---- F2FRansac (src/VISystem.cpp:612-769): lane per iteration, shared normal vectors
__global__ __launch_bounds__(256) void k_f2f(const vis_keypoint* __restrict__ pts1, const vis_keypoint* __restrict__ pts2, int m,
                                             float fx, float fy, float cx, float cy, const float* __restrict__ rot,
                                             const int32_t* __restrict__ sample_idx, int iters, double threshold,
                                             double* __restrict__ nv, float* __restrict__ counts) {
    // phase 1 (grid-stride over points) is done by a first launch with iters == 0
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (iters == 0) {
        if (gid >= m) return;
        const float u1 = pts1[gid].x, v1 = pts1[gid].y, u2 = pts2[gid].x, v2 = pts2[gid].y;
        double a[3] = {(double)((u1 - cx) / fx), (double)((v1 - cy) / fy), 1.0};
        double b[3] = {(double)((u2 - cx) / fx), (double)((v2 - cy) / fy), 1.0};
        const double na = sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
        const double nb = sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
        for (int k = 0; k < 3; k++) { a[k] /= na; b[k] /= nb; }
        double Rm[9]; for (int i = 0; i < 9; i++) Rm[i] = (double)rot[i];
        const double rb[3] = {(Rm[0] * b[0] + Rm[1] * b[1]) + Rm[2] * b[2], (Rm[3] * b[0] + Rm[4] * b[1]) + Rm[5] * b[2],
                              (Rm[6] * b[0] + Rm[7] * b[1]) + Rm[8] * b[2]};
        cross3(a, rb, nv + 3 * (size_t)gid);
        return;
    }
    if (gid >= iters) return;
    const int i1 = sample_idx[2 * gid], i2 = sample_idx[2 * gid + 1];
    double d[3]; cross3(nv + 3 * (size_t)i1, nv + 3 * (size_t)i2, d);
    float count = -1.f;                                            // -1: degenerate sample (skipped by the reference)
    if (d[0] != 0.0 || d[1] != 0.0 || d[2] != 0.0) {
        const double dn = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
        for (int k = 0; k < 3; k++) d[k] /= dn;
        count = 0.f;
        for (int i = 0; i < m; i++) {
            const double error = -1000.0 / log10(fabs(dot3(d, nv + 3 * (size_t)i)));
            if (error < threshold) count += 1.f;
        }
    }
    counts[4 * (size_t)gid] = count;
    counts[4 * (size_t)gid + 1] = (float)d[0]; counts[4 * (size_t)gid + 2] = (float)d[1]; counts[4 * (size_t)gid + 3] = (float)d[2];
}

// ------------------------------------------------------------------------------------------------
static PoseParams make_pose_params(const vis_ctx* ctx, int max_iters, int mcap) {
    PoseParams P;
    P.fx_inv = 1. / ctx->p.fx; P.cx = ctx->p.cx; P.cy = ctx->p.cy;
    P.thr = ctx->p.ransac_threshold / ctx->p.fx;        // findEssentialMat: threshold /= focal
    P.prob = ctx->p.ransac_prob; P.seed = ctx->p.ransac_seed;
    P.max_iters = max_iters; P.adaptive = ctx->p.ransac_adaptive; P.mcap = mcap;
    P.sample_table = (max_iters == ctx->sample_iters && ctx->sample_seed == ctx->p.ransac_seed) ? ctx->d_sample_table : nullptr;
    P.table_max_m = ctx->sample_max_m;
    return P;
}

struct PoseGrids { int hyp_list, roots, models, score; };
static PoseGrids pose_grids(int device) {
    hipDeviceProp_t pr; int cus = 256;
    if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
    auto occ = [&](const void* k, int block, size_t lds, int fallback) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, block, lds) != hipSuccess || n < 1) { (void)hipGetLastError(); n = fallback; }
        return n * cus;
    };
    PoseGrids g;
    g.hyp_list = occ((const void*)k_ransac_hyp_list, 64, HYP_LDS_BYTES, 8) * 2;     // equal items: two rounds
    g.roots = occ((const void*)k_hyp_roots_packed, 64, 0, 8);
    g.models = occ((const void*)k_hyp_models, 256, 0, 4) * 4;
    g.score = occ((const void*)k_hyp_score, 256, 0, 4);
    return g;
}

// generic driver used by the batch path and the host-pointer entry points.
// d_p1/d_p2: npairs x mcap x 2 floats; d_npts: npairs
int pose_run(vis_ctx* ctx, int npairs, int mcap, int max_iters, const float* d_p1, const float* d_p2, const int32_t* d_npts,
             double* d_n1, double* d_n2, int32_t* d_samples, double* d_models, int32_t* d_counts, int32_t* d_rstate,
             const double* d_E_in, uint8_t* d_mask, PoseOut* d_pose, int do_ransac, int do_pose, int32_t* d_worklist, double* d_hyp) {
    hipStream_t st = ctx->stream;
    PoseParams P = make_pose_params(ctx, max_iters, mcap);
    if (!ctx->pose_attr_set) {       // the > 64 KiB dynamic-LDS opt-in is stored per device function: once per context (= per device)
        HIPCHK(ctx, hipFuncSetAttribute((const void*)k_ransac_hyp, hipFuncAttributeMaxDynamicSharedMemorySize, HYP_LDS_BYTES));
        HIPCHK(ctx, hipFuncSetAttribute((const void*)k_ransac_hyp_list, hipFuncAttributeMaxDynamicSharedMemorySize, HYP_LDS_BYTES));
        ctx->pose_attr_set = true;
    }
    hipLaunchKernelGGL(k_pose_prep, dim3(npairs), dim3(256), 0, st, P, d_p1, d_p2, d_npts, d_n1, d_n2, d_samples, d_rstate, do_ransac ? d_worklist : (int32_t*)nullptr);
    if (do_ransac) {
        // per chunk of hypotheses: (A) minimal solver up to the degree-10 polynomial, four lanes per hypothesis, 14.5 KB LDS per wave;
        // (B) its real roots, 16 lanes per hypothesis; (C) models + inlier counts, 256 threads per 16 hypotheses;
        // then the sequential accept/adaptive-bound rule is replayed by k_ransac_scan.
#ifdef VIS_AB_KNOBS       // diagnostic build only (make EXTRA=-DVIS_AB_KNOBS): the shipped library reads no environment variable
        static const int first_chunk = getenv("VIS_RANSAC_FIRST") ? std::max(4, std::min(16, atoi(getenv("VIS_RANSAC_FIRST")) & ~3)) : 16;
        static const bool roots16 = getenv("VIS_ROOTS_16LANE") != nullptr;      // the 16-lanes-per-polynomial kernel for every chunk
#else
        // 16: the headline's degenerate pairs stop within 4 hypotheses and would be as fast with 8 or 4, but S-752P's pairs need 8.7 on
        // average and every pair past the first chunk costs a whole 64-hypothesis item of the list kernels (same-box A/B 16 / 8 / 4,
        // tools/r5_first.sh: S-752 404 / 406 / 405 k frames/s, S-752P 386 / 342 / 298 k).  One value for every batch size: n_models (the
        // work counter of the pose record) depends on it, and a stream's records must not depend on how it is cut into batches.
        const int first_chunk = 16;
        const bool roots16 = false;
#endif
        // adaptive runs: the first chunk of hypotheses of every pair, then only the pairs whose bound is still above it (work list): the
        // sequential accept / adaptive-bound rule is replayed over the same hypothesis sequence, so the chunking never changes a result.
        // With the adaptive stop off every pair needs every hypothesis: no first chunk, the first scan (hi = 0) only builds the work list.
        const int first = ctx->p.ransac_adaptive ? std::min(first_chunk, std::max(max_iters, 1)) : 0;
        const size_t S = (size_t)npairs * max_iters;
        if (first > 0) {
            hipLaunchKernelGGL(k_ransac_hyp, dim3((first + QH - 1) / QH, npairs), dim3(64), HYP_LDS_BYTES, st, P, 0, first, npairs, d_n1, d_n2,
                               d_samples, d_rstate, d_hyp, S);
            // 16 lanes per polynomial for a handful of pairs (a single call's latency), one polynomial per lane for a batch: a
            // third of the vector instructions, and the longer chain of the pose stream is hidden behind the detect chain (+0.5 %)
            if (npairs >= 64)
                hipLaunchKernelGGL(k_hyp_roots_packed, dim3((npairs + 3) / 4), dim3(64), 0, st, P, 0, first, npairs, d_rstate, d_hyp, S, (const int32_t*)nullptr, 0);
            else
                hipLaunchKernelGGL(k_hyp_roots, dim3(npairs), dim3(256), 0, st, P, 0, first, npairs, d_rstate, d_hyp, S, (const int32_t*)nullptr, 0);
            hipLaunchKernelGGL(k_hyp_models, dim3((npairs + 15) / 16), dim3(256), 0, st, P, 0, first, npairs, d_rstate, d_hyp, S, d_models, (const int32_t*)nullptr, 0);
            hipLaunchKernelGGL(k_hyp_score, dim3(npairs), dim3(256), 0, st, P, 0, first, npairs, d_rstate, d_n1, d_n2, d_hyp, S, d_models,
                               d_counts, (const int32_t*)nullptr, 0);
        }
        hipLaunchKernelGGL(k_ransac_scan, dim3(npairs), dim3(64), 0, st, P, first, d_counts, d_rstate, d_worklist);
        if (max_iters > first) {
            const int chunks = (max_iters - first + 63) / 64;
            // grids = what is resident on the chip at once (occupancy x CUs), walked in strides / by claiming: one workgroup per possible
            // item floods the adaptive headline, whose list is almost empty, with empty workgroups (65 k of them: - 5 % frames/s)
            if (!ctx->pose_grids_set) { const PoseGrids g_ = pose_grids(ctx->device); ctx->pose_grid[0] = g_.hyp_list; ctx->pose_grid[1] = g_.roots; ctx->pose_grid[2] = g_.models; ctx->pose_grid[3] = g_.score; ctx->pose_grids_set = true; }
            const PoseGrids G = {ctx->pose_grid[0], ctx->pose_grid[1], ctx->pose_grid[2], ctx->pose_grid[3]};      // per context = per device, behind its own LDS opt-in (not a process-wide static)
            const int nb = (int)std::min<long long>(G.hyp_list, (long long)npairs * chunks * (64 / QH));
            // k_hyp_score: a workgroup per sub-item when every pair is on the list (adaptive stop off: known on the host) -- workgroups that
            // walk equal items in strides stay in lockstep, their load phases (models, four rows of points per round) coincide on a CU
            // and nothing computes meanwhile: config 3 ms_pose 2.92 / 2.83 / 2.76 / 2.70 / 2.68 at 1024 / 2048 / 4096 / 8192 / 16384
            // workgroups for 16256 sub-items.  With the adaptive stop the list is short and not known here: twice what is resident.
            const int nsub = (int)std::min<long long>(ctx->p.ransac_adaptive ? 2 * G.score : (1 << 20), (long long)npairs * chunks * 4);   // one workgroup per sub-item (a fixed grid of 2048 walking them: config 3 + 0.17 ms)
            hipLaunchKernelGGL(k_ransac_hyp_list, dim3(nb), dim3(64), HYP_LDS_BYTES, st, P, first, max_iters,
                               npairs, d_n1, d_n2, d_samples, d_rstate, d_hyp, S, (const int32_t*)d_worklist, chunks);
            if (roots16)
                hipLaunchKernelGGL(k_hyp_roots, dim3(nsub), dim3(256), 0, st, P, first, max_iters, npairs, d_rstate, d_hyp, S,
                                   (const int32_t*)d_worklist, chunks);
            else           // one hypothesis per lane
                // one workgroup (wave) per 64-hypothesis item: the items differ in their numbers of real roots, a fixed grid walking
                // them in strides left a quarter of the chip idle at the end (2.0 -> 1.5 ms per 1.02 M polynomials)
                hipLaunchKernelGGL(k_hyp_roots_packed, dim3(std::min(G.roots, npairs * chunks)), dim3(64), 0, st, P, first, max_iters, npairs, d_rstate,
                                   d_hyp, S, (const int32_t*)d_worklist, chunks);
            hipLaunchKernelGGL(k_hyp_models, dim3(std::min(G.models, (npairs * chunks + 3) / 4)), dim3(256), 0, st, P, first, max_iters, npairs, d_rstate,
                               d_hyp, S, d_models, (const int32_t*)d_worklist, chunks);
            hipLaunchKernelGGL(k_hyp_score, dim3(nsub), dim3(256), 0, st, P, first, max_iters, npairs, d_rstate, d_n1, d_n2, d_hyp, S,
                               d_models, d_counts, (const int32_t*)d_worklist, chunks);
            hipLaunchKernelGGL(k_ransac_scan, dim3(npairs), dim3(64), 0, st, P, max_iters, d_counts, d_rstate, (int32_t*)nullptr);
        }
    }
    const int nsplit = std::max(1, std::min(32, (4 * mcap + 4095) / 4096));       // ~16 triangulations per thread
    if (do_pose) hipLaunchKernelGGL(k_pose_svd, dim3((npairs + 63) / 64), dim3(64), 0, st, P, (const double*)d_models, (const int32_t*)d_rstate, d_E_in, d_hyp, npairs);
    hipLaunchKernelGGL(k_pose_final, dim3(npairs, nsplit), dim3(256), 0, st, P, d_n1, d_n2, d_models, d_rstate, d_E_in, d_mask, d_pose, do_pose, (const double*)d_hyp);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}

int f2f_run(vis_ctx* ctx, const vis_keypoint* d_pts1, const vis_keypoint* d_pts2, int m, const float* d_rot,
            const int32_t* d_idx, int iters, double* d_nv, float* d_counts) {
    hipStream_t st = ctx->stream;
    const float fx = (float)ctx->p.fx, fy = (float)ctx->p.fy, cx = (float)ctx->p.cx, cy = (float)ctx->p.cy;
    hipLaunchKernelGGL(k_f2f, dim3((m + 255) / 256), dim3(256), 0, st, d_pts1, d_pts2, m, fx, fy, cx, cy, d_rot, d_idx, 0,
                       ctx->p.f2f_threshold, d_nv, d_counts);
    if (iters > 0)
        hipLaunchKernelGGL(k_f2f, dim3((iters + 255) / 256), dim3(256), 0, st, d_pts1, d_pts2, m, fx, fy, cx, cy, d_rot, d_idx, iters,
                           ctx->p.f2f_threshold, d_nv, d_counts);
    HIPCHK(ctx, hipGetLastError());
    return VIS_OK;
}
