// oracle/pose.cpp -- CPU ORACLE (test infrastructure only; see vis_oracle.h header).
//
// Restates the pose inner loops the reference reaches through OpenCV calib3d:
//   findEssentialMat(p1,p2,focal,pp,RANSAC,0.999,1.0)   /root/reference/src/VISystem.cpp:1679-1680
//   recoverPose(E,p1,p2,R,t,focal,pp)                   /root/reference/src/VISystem.cpp:1701
// and the reference's own VISystem::F2FRansac            /root/reference/src/VISystem.cpp:612-769
//
// PARITY UNPINNED vs OpenCV: calib3d's five-point.cpp carries machine-generated coefficient
// expansions and a Durand-Kerner root finder that cannot be reproduced bit for bit without its
// source.  This file implements the same published algorithm (Nister 2004: null space of the 5x9
// epipolar system, 10 cubic constraints, Gauss-Jordan to a 3x3 polynomial matrix B(z), det B = 0 as a
// degree-10 polynomial, back-substitution) with a documented, deterministic realisation:
//   null space  : Householder QR of Q^T (orthonormal basis X,Y,Z,W)
//   real roots  : derivative-interlacing bisection (Rolle), roots taken in ascending order
//   (x,y) from z: largest-norm row cross product of B(z)
// RANSAC (cv::RANSACPointSetRegistrator::run semantics, cv::RNG sample stream, adaptive iteration
// count, float Sampson error vs float threshold) and recoverPose (4 candidates, DLT triangulation,
// cheirality with the 50-unit distance cut) follow SURVEY.md Appendix A.3.
#include "vis_oracle.h"
#include "oracle_internal.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------- small linear algebra
// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major), n <= 4.
// A is destroyed (diagonal = eigenvalues), V columns = eigenvectors.
static void jacobi_eig(int n, double* A, double* V) {
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                double app = A[p * n + p], aqq = A[q * n + q];
                double theta = (aqq - app) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; k++) {                 // A <- A J
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {                 // A <- J^T A
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
}

static inline void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ---------------------------------------------------------------------------- monomial tables
// variables 0:x 1:y 2:z 3:1.  quadratic monomials (10) and cubic monomials (20, elimination order).
static const int QUAD_EXP[10][3] = {{2,0,0},{1,1,0},{1,0,1},{0,2,0},{0,1,1},{0,0,2},{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
static const int CUB_EXP[20][3] = {
    {3,0,0},{0,3,0},{2,1,0},{1,2,0},{2,0,1},{2,0,0},{0,2,1},{0,2,0},{1,1,1},{1,1,0},
    {1,0,2},{1,0,1},{1,0,0},{0,1,2},{0,1,1},{0,1,0},{0,0,3},{0,0,2},{0,0,1},{0,0,0}};
struct Tables {
    int q_of[4][4];      // lin var i * lin var j -> quad index
    int c_of[10][4];     // quad monomial * lin var -> cubic index
    Tables() {
        auto find = [](const int (*tab)[3], int n, int a, int b, int c) {
            for (int i = 0; i < n; i++) if (tab[i][0] == a && tab[i][1] == b && tab[i][2] == c) return i;
            return -1;
        };
        int ve[4][3] = {{1,0,0},{0,1,0},{0,0,1},{0,0,0}};
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++)
            q_of[i][j] = find(QUAD_EXP, 10, ve[i][0] + ve[j][0], ve[i][1] + ve[j][1], ve[i][2] + ve[j][2]);
        for (int m = 0; m < 10; m++) for (int v = 0; v < 4; v++)
            c_of[m][v] = find(CUB_EXP, 20, QUAD_EXP[m][0] + ve[v][0], QUAD_EXP[m][1] + ve[v][1], QUAD_EXP[m][2] + ve[v][2]);
    }
};
static const Tables TB;

static void mul_ll(const double* a, const double* b, double* q) {        // lin*lin -> quad (accumulate)
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) q[TB.q_of[i][j]] += a[i] * b[j];
}
static void mul_ql(const double* q, const double* l, double* c, double sgn) {   // quad*lin -> cubic (accumulate)
    for (int m = 0; m < 10; m++) for (int v = 0; v < 4; v++) c[TB.c_of[m][v]] += sgn * (q[m] * l[v]);
}

// ---------------------------------------------------------------------------- polynomial helpers
static double poly_eval(const double* c, int deg, double x) {
    double r = c[deg];
    for (int i = deg - 1; i >= 0; i--) r = r * x + c[i];
    return r;
}

// Real roots of c[0..deg] in ascending order (returns count <= deg).
// Method ("derivative interlacing"): by Rolle's theorem the roots of q' separate the roots of q, so the
// real roots of the k-th derivative tower p^(deg-1), p^(deg-2), ..., p are found level by level: at each
// level the previous level's roots (plus the Cauchy bound +-B) cut the line into intervals that hold at
// most one root each; an interval with a sign change is bisected.  Inner levels only need to separate
// (40 halvings); the last level bisects until the interval cannot shrink in double precision.
// Control flow is data-independent apart from the sign tests -- the HIP kernel runs the same steps on 64
// hypotheses in lock step.  Even-multiplicity roots (no sign change) are not reported.
static const int BISECT_INNER = 40, BISECT_FINAL = 200;

static int real_roots(const double* cin, int deg, double* roots) {
    double c[11]; double mx = 0;
    for (int i = 0; i <= deg; i++) mx = std::max(mx, std::fabs(cin[i]));
    if (mx == 0) return 0;
    for (int i = 0; i <= deg; i++) c[i] = cin[i] / mx;
    while (deg > 0 && std::fabs(c[deg]) < 1e-15) deg--;
    if (deg == 0) return 0;
    double B = 0;
    for (int i = 0; i < deg; i++) B = std::max(B, std::fabs(c[i] / c[deg]));
    B += 1.0;
    double prev[11]; int nprev = 0;
    for (int d = 1; d <= deg; d++) {
        const int k = deg - d;                       // q = k-th derivative of p, degree d
        double q[11];
        for (int i = 0; i <= d; i++) {
            double f = 1.0;
            for (int j = 0; j < k; j++) f *= (double)(i + k - j);
            q[i] = c[i + k] * f;
        }
        double cur[11]; int ncur = 0;
        for (int j = 0; j <= nprev; j++) {
            double lo = j == 0 ? -B : prev[j - 1];
            double hi = j == nprev ? B : prev[j];
            double flo = poly_eval(q, d, lo), fhi = poly_eval(q, d, hi);
            if ((flo < 0) == (fhi < 0)) continue;    // no sign change: no (odd) root here
            const int nit = d == deg ? BISECT_FINAL : BISECT_INNER;
            for (int it = 0; it < nit; it++) {
                const double m = 0.5 * (lo + hi);
                if (m <= lo || m >= hi) break;
                const double fm = poly_eval(q, d, m);
                if ((fm < 0) == (flo < 0)) lo = m; else hi = m;
            }
            cur[ncur++] = 0.5 * (lo + hi);
        }
        nprev = ncur;
        for (int j = 0; j < ncur; j++) prev[j] = cur[j];
    }
    for (int j = 0; j < nprev; j++) roots[j] = prev[j];
    return nprev;
}

// ---------------------------------------------------------------------------- five-point solver
// q1xy/q2xy: 5 x 2 normalised image points.  Es: up to 10 matrices, row-major, x2^T E x1 = 0.
static int five_point_ex(const double* q1, const double* q2, double* Es, double* poly_out, double* roots_out, int* nroots_out);
int five_point(const double* q1, const double* q2, double* Es) { return five_point_ex(q1, q2, Es, nullptr, nullptr, nullptr); }
// poly_out (11 ascending coefficients of det B(z)), roots_out / nroots_out (its real roots as found by real_roots): test hooks
static int five_point_ex(const double* q1, const double* q2, double* Es, double* poly_out, double* roots_out, int* nroots_out) {
    if (nroots_out) *nroots_out = 0;
    // A = Q^T (9 x 5), Q row i = [x2x1, x2y1, x2, y2x1, y2y1, y2, x1, y1, 1]
    double A[9][5];
    for (int i = 0; i < 5; i++) {
        double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int k = 0; k < 9; k++) A[k][i] = r[k];
    }
    // Householder QR of A; keep the reflectors
    double vs[5][9]; double betas[5];
    for (int k = 0; k < 5; k++) {
        double nrm = 0;
        for (int i = k; i < 9; i++) nrm += A[i][k] * A[i][k];
        nrm = std::sqrt(nrm);
        for (int i = 0; i < 9; i++) vs[k][i] = 0;
        if (nrm < 1e-300) { betas[k] = 0; continue; }
        double alpha = A[k][k] >= 0 ? -nrm : nrm;
        for (int i = k; i < 9; i++) vs[k][i] = A[i][k];
        vs[k][k] -= alpha;
        double vn = 0; for (int i = k; i < 9; i++) vn += vs[k][i] * vs[k][i];
        if (vn < 1e-300) { betas[k] = 0; continue; }
        betas[k] = 2.0 / vn;
        for (int j = k; j < 5; j++) {
            double d = 0; for (int i = k; i < 9; i++) d += vs[k][i] * A[i][j];
            d *= betas[k];
            for (int i = k; i < 9; i++) A[i][j] -= d * vs[k][i];
        }
    }
    // null-space basis: columns 5..8 of H0 H1 .. H4
    double Bs[4][9];
    for (int j = 0; j < 4; j++) {
        double e[9] = {0}; e[5 + j] = 1.0;
        for (int k = 4; k >= 0; k--) {
            double d = 0; for (int i = k; i < 9; i++) d += vs[k][i] * e[i];
            d *= betas[k];
            for (int i = k; i < 9; i++) e[i] -= d * vs[k][i];
        }
        for (int i = 0; i < 9; i++) Bs[j][i] = e[i];
    }
    // E entries as linear polynomials in (x,y,z,1)
    double El[3][3][4];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) for (int v = 0; v < 4; v++) El[r][c][v] = Bs[v][3 * r + c];
    // constraint matrix M (10 x 20)
    double M[10][20];
    for (int r = 0; r < 10; r++) for (int c = 0; c < 20; c++) M[r][c] = 0;
    {   // row 0: det(E)
        double q[10];
        auto det_term = [&](int a, int b, int c, int d, int e_, int f, double sgn) {
            // sgn * E[0][a] * (E[1][b]*E[2][c] - E[1][d]*E[2][e_])  (f unused)
            (void)f;
            for (int i = 0; i < 10; i++) q[i] = 0;
            mul_ll(El[1][b], El[2][c], q);
            double q2[10]; for (int i = 0; i < 10; i++) q2[i] = 0;
            mul_ll(El[1][d], El[2][e_], q2);
            for (int i = 0; i < 10; i++) q[i] -= q2[i];
            mul_ql(q, El[0][a], M[0], sgn);
        };
        det_term(0, 1, 2, 2, 1, 0, 1.0);
        det_term(1, 0, 2, 2, 0, 0, -1.0);
        det_term(2, 0, 1, 1, 0, 0, 1.0);
    }
    {   // rows 1..9: (E E^T - 0.5 tr(E E^T) I) E
        double EEt[3][3][10];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            for (int m = 0; m < 10; m++) EEt[i][j][m] = 0;
            for (int k = 0; k < 3; k++) mul_ll(El[i][k], El[j][k], EEt[i][j]);
        }
        double tr[10];
        for (int m = 0; m < 10; m++) tr[m] = 0.5 * ((EEt[0][0][m] + EEt[1][1][m]) + EEt[2][2][m]);
        for (int i = 0; i < 3; i++) for (int m = 0; m < 10; m++) EEt[i][i][m] -= tr[m];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) mul_ql(EEt[i][k], El[k][j], M[1 + 3 * i + j], 1.0);
    }
    // Gauss-Jordan with partial pivoting on the left 10 x 10 block
    for (int col = 0; col < 10; col++) {
        int piv = col; double best = std::fabs(M[col][col]);
        for (int r = col + 1; r < 10; r++) if (std::fabs(M[r][col]) > best) { best = std::fabs(M[r][col]); piv = r; }
        if (best < 1e-300) return 0;
        if (piv != col) for (int c = 0; c < 20; c++) std::swap(M[piv][c], M[col][c]);
        double inv = 1.0 / M[col][col];
        for (int c = col; c < 20; c++) M[col][c] *= inv;
        for (int r = 0; r < 10; r++) {
            if (r == col) continue;
            double f = M[r][col];
            if (f == 0.0) continue;
            for (int c = col; c < 20; c++) M[r][c] -= f * M[col][c];
        }
    }
    // B(z): rows from (4,5), (6,7), (8,9): k = row_a - z*row_b
    // column blocks of the right part: [xz^2,xz,x | yz^2,yz,y | z^3,z^2,z,1] = cols 10..19
    double Bx[3][4], By[3][4], B1[3][5];            // ascending powers of z
    for (int i = 0; i < 3; i++) {
        const double* a = &M[4 + 2 * i][10];
        const double* b = &M[5 + 2 * i][10];
        // x block: a: z^2,z,1 -> a[0],a[1],a[2] ; z*b: z^3,z^2,z -> b[0],b[1],b[2]
        Bx[i][0] = a[2];            Bx[i][1] = a[1] - b[2]; Bx[i][2] = a[0] - b[1]; Bx[i][3] = -b[0];
        By[i][0] = a[5];            By[i][1] = a[4] - b[5]; By[i][2] = a[3] - b[4]; By[i][3] = -b[3];
        B1[i][0] = a[9];            B1[i][1] = a[8] - b[9]; B1[i][2] = a[7] - b[8]; B1[i][3] = a[6] - b[7]; B1[i][4] = -b[6];
    }
    // det B(z): degree 10
    auto pmul = [](const double* a, int da, const double* b, int db, double* o) {
        for (int i = 0; i <= da + db; i++) o[i] = 0;
        for (int i = 0; i <= da; i++) for (int j = 0; j <= db; j++) o[i + j] += a[i] * b[j];
    };
    double c10[11]; for (int i = 0; i <= 10; i++) c10[i] = 0;
    {
        double t1[8], t2[8], m[8], o[11];
        // + Bx0 * (By1*B12 - B11*By2)
        pmul(By[1], 3, B1[2], 4, t1); pmul(B1[1], 4, By[2], 3, t2);
        for (int i = 0; i <= 7; i++) m[i] = t1[i] - t2[i];
        pmul(Bx[0], 3, m, 7, o); for (int i = 0; i <= 10; i++) c10[i] += o[i];
        // - By0 * (Bx1*B12 - B11*Bx2)
        pmul(Bx[1], 3, B1[2], 4, t1); pmul(B1[1], 4, Bx[2], 3, t2);
        for (int i = 0; i <= 7; i++) m[i] = t1[i] - t2[i];
        pmul(By[0], 3, m, 7, o); for (int i = 0; i <= 10; i++) c10[i] -= o[i];
        // + B10 * (Bx1*By2 - By1*Bx2)
        double u1[7], u2[7], mm[7];
        pmul(Bx[1], 3, By[2], 3, u1); pmul(By[1], 3, Bx[2], 3, u2);
        for (int i = 0; i <= 6; i++) mm[i] = u1[i] - u2[i];
        pmul(B1[0], 4, mm, 6, o); for (int i = 0; i <= 10; i++) c10[i] += o[i];
    }
    double roots[10];
    int nr = real_roots(c10, 10, roots);
    if (poly_out) for (int i = 0; i <= 10; i++) poly_out[i] = c10[i];
    if (roots_out) for (int i = 0; i < nr; i++) roots_out[i] = roots[i];
    if (nroots_out) *nroots_out = nr;
    int count = 0;
    for (int ri = 0; ri < nr && count < 10; ri++) {
        double z = roots[ri];
        double Bz[3][3];
        for (int i = 0; i < 3; i++) {
            Bz[i][0] = poly_eval(Bx[i], 3, z);
            Bz[i][1] = poly_eval(By[i], 3, z);
            Bz[i][2] = poly_eval(B1[i], 4, z);
        }
        double c01[3], c02[3], c12[3];
        cross3(Bz[0], Bz[1], c01); cross3(Bz[0], Bz[2], c02); cross3(Bz[1], Bz[2], c12);
        double n01 = dot3(c01, c01), n02 = dot3(c02, c02), n12 = dot3(c12, c12);
        const double* nv = c01; double nn = n01;
        if (n02 > nn) { nv = c02; nn = n02; }
        if (n12 > nn) { nv = c12; nn = n12; }
        if (!(nn > 0)) continue;
        double inv = 1.0 / std::sqrt(nn);
        double w = nv[2] * inv;
        if (std::fabs(w) < 1e-10) continue;
        double x = (nv[0] * inv) / w, y = (nv[1] * inv) / w;
        double E[9]; double fn = 0;
        for (int i = 0; i < 9; i++) {
            E[i] = ((x * Bs[0][i] + y * Bs[1][i]) + z * Bs[2][i]) + Bs[3][i];
            fn += E[i] * E[i];
        }
        fn = std::sqrt(fn);
        if (!(fn > 0)) continue;
        for (int i = 0; i < 9; i++) Es[9 * count + i] = E[i] / fn;
        count++;
    }
    return count;
}

// ---------------------------------------------------------------------------- RANSAC driver
struct CvRng {                     // cv::RNG (core.hpp): MWC, CV_RNG_COEFF = 4164903690U
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// getSubset (ptsetreg.cpp): 5 distinct indices, redraw on duplicates
static void draw_subset(CvRng& rng, int count, int* idx) {
    for (int i = 0; i < 5; i++) {
        for (;;) {
            int v = idx[i] = rng.uniform(0, count);
            int j = 0;
            for (; j < i; j++) if (v == idx[j]) break;
            if (j == i) break;
        }
    }
}

static int update_num_iters(double p, double ep, int modelPoints, int maxIters) {   // RANSACUpdateNumIters
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, modelPoints);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)std::lrint(num / denom);
}

// EMEstimatorCallback::computeError + findInliers: float Sampson error <= (float)(thr^2)
static int count_inliers(const double* E, const double* n1, const double* n2, int m, double thr, uint8_t* mask) {
    float t = (float)(thr * thr);
    int good = 0;
    for (int i = 0; i < m; i++) {
        double x1 = n1[2 * i], y1 = n1[2 * i + 1], x2 = n2[2 * i], y2 = n2[2 * i + 1];
        double Ex0 = (E[0] * x1 + E[1] * y1) + E[2];
        double Ex1 = (E[3] * x1 + E[4] * y1) + E[5];
        double Ex2 = (E[6] * x1 + E[7] * y1) + E[8];
        double Et0 = (E[0] * x2 + E[3] * y2) + E[6];
        double Et1 = (E[1] * x2 + E[4] * y2) + E[7];
        double x2tEx1 = (x2 * Ex0 + y2 * Ex1) + Ex2;
        double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
        float err = (float)(x2tEx1 * x2tEx1 / (((a + b) + c) + d));
        int f = err <= t;
        if (mask) mask[i] = (uint8_t)f;
        good += f;
    }
    return good;
}

static void normalise_points(const vis_params& p, const float* pxy, int m, std::vector<double>& out) {
    // findEssentialMat(focal, pp): K = [focal 0 cx; 0 focal cy]; (u - cx) / focal via Mat scale 1./f
    out.resize(2 * (size_t)m);
    double inv = 1. / p.fx;
    for (int i = 0; i < m; i++) {
        out[2 * i] = ((double)pxy[2 * i] - p.cx) * inv;
        out[2 * i + 1] = ((double)pxy[2 * i + 1] - p.cy) * inv;
    }
}

int essential_ransac(const vis_params& p, const float* p1xy, const float* p2xy, int m,
                     double E[9], uint8_t* mask_out, int* n_inl, int* iters_run) {
    for (int i = 0; i < 9; i++) E[i] = 0;
    if (n_inl) *n_inl = 0;
    if (iters_run) *iters_run = 0;
    if (mask_out) std::memset(mask_out, 0, (size_t)std::max(m, 0));
    if (m < 5) return VIS_OK;                    // run(): count < modelPoints -> false, empty E
    std::vector<double> n1, n2;
    normalise_points(p, p1xy, m, n1); normalise_points(p, p2xy, m, n2);
    const double thr = p.ransac_threshold / p.fx;   // threshold /= (fx+fy)/2 with fx == fy == focal
    double Es[90];
    if (m == 5) {                                // count == modelPoints: single runKernel, all-ones mask
        int nm = five_point(n1.data(), n2.data(), Es);
        if (nm <= 0) return VIS_OK;
        std::memcpy(E, Es, 9 * sizeof(double));  // SPEC: first model (OpenCV would stack all of them)
        if (mask_out) std::memset(mask_out, 1, 5);
        if (n_inl) *n_inl = 5;
        if (iters_run) *iters_run = 1;
        return VIS_OK;
    }
    CvRng rng(p.ransac_seed);
    int niters = std::max(p.ransac_max_iters, 1), maxGood = 0, iter = 0;
    std::vector<uint8_t> mask(m), best(m, 0);
    for (iter = 0; iter < niters; iter++) {
        int idx[5]; draw_subset(rng, m, idx);
        double s1[10], s2[10];
        for (int k = 0; k < 5; k++) {
            s1[2 * k] = n1[2 * idx[k]]; s1[2 * k + 1] = n1[2 * idx[k] + 1];
            s2[2 * k] = n2[2 * idx[k]]; s2[2 * k + 1] = n2[2 * idx[k] + 1];
        }
        int nm = five_point(s1, s2, Es);
        for (int i = 0; i < nm; i++) {
            int good = count_inliers(Es + 9 * i, n1.data(), n2.data(), m, thr, mask.data());
            if (good > std::max(maxGood, 4)) {
                best = mask; std::memcpy(E, Es + 9 * i, 9 * sizeof(double));
                maxGood = good;
                if (p.ransac_adaptive)
                    niters = update_num_iters(p.ransac_prob, (double)(m - good) / m, 5, niters);
            }
        }
    }
    if (mask_out && maxGood > 0 && m > 0) std::memcpy(mask_out, best.data(), m);
    if (n_inl) *n_inl = maxGood;
    if (iters_run) *iters_run = iter;
    return VIS_OK;
}

// ---------------------------------------------------------------------------- recoverPose
static void svd3_decompose(const double* E, double* U, double* Vt) {
    // eigen of E^T E -> V (sorted by descending eigenvalue), right-handed; U from E v_i
    double A[9], V[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += E[3 * k + i] * E[3 * k + j];
        A[3 * i + j] = s;
    }
    jacobi_eig(3, A, V);
    int ord[3] = {0, 1, 2};
    std::sort(ord, ord + 3, [&](int a, int b) { if (A[4 * a] != A[4 * b]) return A[4 * a] > A[4 * b]; return a < b; });
    double v0[3], v1[3], v2[3];
    for (int k = 0; k < 3; k++) { v0[k] = V[3 * k + ord[0]]; v1[k] = V[3 * k + ord[1]]; }
    cross3(v0, v1, v2);
    double u0[3], u1[3], u2[3];
    for (int r = 0; r < 3; r++) { u0[r] = dot3(E + 3 * r, v0); u1[r] = dot3(E + 3 * r, v1); }
    double n0 = std::sqrt(dot3(u0, u0));
    for (int r = 0; r < 3; r++) u0[r] /= n0;
    double pr = dot3(u0, u1);
    for (int r = 0; r < 3; r++) u1[r] -= pr * u0[r];
    double nn1 = std::sqrt(dot3(u1, u1));
    for (int r = 0; r < 3; r++) u1[r] /= nn1;
    cross3(u0, u1, u2);
    for (int r = 0; r < 3; r++) { U[3 * r] = u0[r]; U[3 * r + 1] = u1[r]; U[3 * r + 2] = u2[r]; }
    for (int c = 0; c < 3; c++) { Vt[c] = v0[c]; Vt[3 + c] = v1[c]; Vt[6 + c] = v2[c]; }
}

static void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
        C[3 * i + j] = s;
    }
}

// cheirality test of one correspondence under P1 = [R|t] (P0 = [I|0]), dist = 50
static bool cheirality(const double* R, const double* t, double x1, double y1, double x2, double y2) {
    double P[12] = {R[0], R[1], R[2], t[0], R[3], R[4], R[5], t[1], R[6], R[7], R[8], t[2]};
    double A[16];
    // rows: x1*P0[2]-P0[0], y1*P0[2]-P0[1], x2*P[2]-P[0], y2*P[2]-P[1]
    A[0] = -1; A[1] = 0;  A[2] = x1; A[3] = 0;
    A[4] = 0;  A[5] = -1; A[6] = y1; A[7] = 0;
    for (int c = 0; c < 4; c++) { A[8 + c] = x2 * P[8 + c] - P[c]; A[12 + c] = y2 * P[8 + c] - P[4 + c]; }
    double AtA[16], V[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        double s = 0; for (int k = 0; k < 4; k++) s += A[4 * k + i] * A[4 * k + j];
        AtA[4 * i + j] = s;
    }
    jacobi_eig(4, AtA, V);
    int mn = 0; for (int i = 1; i < 4; i++) if (AtA[5 * i] < AtA[5 * mn]) mn = i;
    double X[4] = {V[mn], V[4 + mn], V[8 + mn], V[12 + mn]};
    bool ok = (X[2] * X[3]) > 0;
    double Xn[3] = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};
    ok = ok && (Xn[2] < 50.0);
    double z2 = ((P[8] * Xn[0] + P[9] * Xn[1]) + P[10] * Xn[2]) + P[11];
    ok = ok && (z2 > 0) && (z2 < 50.0);
    return ok;
}

int recover_pose(const vis_params& p, const double* E, const float* p1xy, const float* p2xy, int m,
                 double* R, double* t, int* n_good) {
    std::vector<double> n1, n2;
    normalise_points(p, p1xy, m, n1); normalise_points(p, p2xy, m, n2);
    double U[9], Vt[9]; svd3_decompose(E, U, Vt);
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1}, Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9], R1[9], R2[9];
    mat3_mul(U, W, T); mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T); mat3_mul(T, Vt, R2);
    double tp[3] = {U[2], U[5], U[8]}, tn[3] = {-U[2], -U[5], -U[8]};
    const double* Rs[4] = {R1, R2, R1, R2};
    const double* ts[4] = {tp, tp, tn, tn};
    int good[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < m; i++)
            good[c] += cheirality(Rs[c], ts[c], n1[2 * i], n1[2 * i + 1], n2[2 * i], n2[2 * i + 1]) ? 1 : 0;
    int sel;
    if (good[0] >= good[1] && good[0] >= good[2] && good[0] >= good[3]) sel = 0;
    else if (good[1] >= good[0] && good[1] >= good[2] && good[1] >= good[3]) sel = 1;
    else if (good[2] >= good[0] && good[2] >= good[1] && good[2] >= good[3]) sel = 2;
    else sel = 3;
    std::memcpy(R, Rs[sel], 9 * sizeof(double));
    std::memcpy(t, ts[sel], 3 * sizeof(double));
    if (n_good) *n_good = good[sel];
    return VIS_OK;
}

}  // namespace orc

using namespace orc;

extern "C" int orc_five_point_poly(const double* q1xy, const double* q2xy, double* Es, double* poly11, double* roots10, int* n_roots) {
    if (!q1xy || !q2xy || !Es || !poly11 || !roots10 || !n_roots) return VIS_E_INVALID;
    return orc::five_point_ex(q1xy, q2xy, Es, poly11, roots10, n_roots);
}
extern "C" int orc_five_point(const double* q1xy, const double* q2xy, double* Es) {
    if (!q1xy || !q2xy || !Es) return VIS_E_INVALID;
    return five_point(q1xy, q2xy, Es);
}

extern "C" int orc_ransac_samples(uint64_t seed, int count, int iters, int32_t* idx5) {
    if (count < 5 || !idx5) return VIS_E_INVALID;
    CvRng rng(seed);
    for (int i = 0; i < iters; i++) { int idx[5]; draw_subset(rng, count, idx); for (int k = 0; k < 5; k++) idx5[5 * i + k] = idx[k]; }
    return VIS_OK;
}

extern "C" int orc_essential_ransac(const vis_params* p, const float* p1xy, const float* p2xy, int m,
                                    double E[9], uint8_t* mask, int* n_inliers, int* iters_run) {
    if (!p || m < 0 || (m && (!p1xy || !p2xy)) || !E) return VIS_E_INVALID;
    return essential_ransac(*p, p1xy, p2xy, m, E, mask, n_inliers, iters_run);
}

extern "C" int orc_recover_pose(const vis_params* p, const double E[9], const float* p1xy,
                                const float* p2xy, int m, double R[9], double t[3], int* n_good) {
    if (!p || !E || m < 0 || !R || !t) return VIS_E_INVALID;
    return recover_pose(*p, E, p1xy, p2xy, m, R, t, n_good);
}

// VISystem::F2FRansac, src/VISystem.cpp:612-769.  SPEC: sample indices come from an
// explicit array (the reference draws rand()%(n-1) unseeded, :712-713); m < 2 -> zero vector.  The loop
// indexes the UNSORTED normalVectors (:715) exactly as the reference does; the degeneracy sort (:685-698)
// has no effect on the result and is omitted.
extern "C" int orc_f2f_ransac(const vis_params* p, const vis_keypoint* pts1, const vis_keypoint* pts2, int m,
                              const float rot[9], const int32_t* sample_idx, int iters,
                              float scale, float out_t[3], int* count_max) {
    if (!p || !out_t || m < 0 || iters < 0) return VIS_E_INVALID;
    out_t[0] = out_t[1] = out_t[2] = 0.f;
    if (count_max) *count_max = 0;
    if (m < 2) return VIS_OK;
    float fx = (float)p->fx, fy = (float)p->fy, cx = (float)p->cx, cy = (float)p->cy;
    std::vector<double> nv(3 * (size_t)m);
    double Rm[9]; for (int i = 0; i < 9; i++) Rm[i] = (double)rot[i];
    for (int i = 0; i < m; i++) {
        float u1 = pts1[i].x, v1 = pts1[i].y, u2 = pts2[i].x, v2 = pts2[i].y;
        double a[3] = {(double)((u1 - cx) / fx), (double)((v1 - cy) / fy), 1.0};
        double b[3] = {(double)((u2 - cx) / fx), (double)((v2 - cy) / fy), 1.0};
        double na = std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]);
        double nb = std::sqrt((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]);
        for (int k = 0; k < 3; k++) { a[k] /= na; b[k] /= nb; }
        double rb[3] = {(Rm[0] * b[0] + Rm[1] * b[1]) + Rm[2] * b[2], (Rm[3] * b[0] + Rm[4] * b[1]) + Rm[5] * b[2],
                        (Rm[6] * b[0] + Rm[7] * b[1]) + Rm[8] * b[2]};
        cross3(a, rb, &nv[3 * (size_t)i]);
    }
    float countMax = 0; float best[3] = {0, 0, 0};
    for (int it = 0; it < iters; it++) {
        int i1 = sample_idx[2 * it], i2 = sample_idx[2 * it + 1];
        if (i1 < 0 || i1 >= m || i2 < 0 || i2 >= m) return VIS_E_INVALID;
        double d[3]; cross3(&nv[3 * (size_t)i1], &nv[3 * (size_t)i2], d);
        if (d[0] != 0.0 || d[1] != 0.0 || d[2] != 0.0) {
            double dn = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
            for (int k = 0; k < 3; k++) d[k] /= dn;
            float count = 0;
            for (int i = 0; i < m; i++) {
                double error = -1000.0 / std::log10(std::fabs(dot3(d, &nv[3 * (size_t)i])));
                if (error < p->f2f_threshold) count++;
            }
            if (count > countMax) { countMax = count; for (int k = 0; k < 3; k++) best[k] = (float)d[k]; }
        }
    }
    for (int k = 0; k < 3; k++) out_t[k] = scale * best[k];
    if (count_max) *count_max = (int)countMax;
    return VIS_OK;
}

extern "C" int orc_pipeline_frame(const vis_params* p, const uint8_t* img, int w, int h, int stride,
                                  const vis_keypoint* prev_kps, const uint8_t* prev_desc, int n_prev,
                                  vis_keypoint* kps, uint8_t* desc, int cap, orc_frame_result* res) {
    if (!p || !img || !res) return VIS_E_INVALID;
    std::memset(res, 0, sizeof(*res));
    // Camera::Update (src/Camera.cpp:63-72): the half pyramid is built every frame by the reference
    std::vector<uint8_t> lv[5]; uint8_t* lp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t hw[5], hh[5];
    orc_half_pyramid_dims(w, h, hw, hh);                         // cv::resize's rounded sizes: 375 -> 188, not 375 >> 1
    for (int l = 1; l < 5; l++) { lv[l].resize((size_t)hw[l] * hh[l] + 16); lp[l] = lv[l].data(); }
    orc_half_pyramid(img, w, h, stride, lp);
    std::vector<vis_keypoint> k; std::vector<uint8_t> d;
    int rc = orb_detect_compute(*p, img, w, h, stride, k, d);
    if (rc) return rc;
    res->n_kp = (int)k.size();
    if ((int)k.size() > cap) return VIS_E_CAPACITY;
    if (kps && !k.empty()) std::memcpy(kps, k.data(), k.size() * sizeof(vis_keypoint));
    if (desc && !d.empty()) std::memcpy(desc, d.data(), d.size());
    if (!prev_kps || n_prev <= 0 || k.empty()) return VIS_OK;
    std::vector<vis_dmatch> k12(2 * (size_t)n_prev), k21(2 * k.size());
    knn2_hamming(prev_desc, n_prev, d.data(), (int)k.size(), k12.data());   // both directions, like
    knn2_hamming(d.data(), (int)k.size(), prev_desc, n_prev, k21.data());   // src/Matcher.cpp:86,88
    std::vector<vis_dmatch> sym, good;
    good_matches(*p, prev_kps, n_prev, k.data(), (int)k.size(), k12.data(), k21.data(), sym, good);
    res->n_sym = (int)sym.size(); res->n_good = (int)good.size();
    std::vector<float> a(2 * good.size()), b(2 * good.size());
    for (size_t i = 0; i < good.size(); i++) {
        a[2 * i] = prev_kps[good[i].queryIdx].x; a[2 * i + 1] = prev_kps[good[i].queryIdx].y;
        b[2 * i] = k[good[i].trainIdx].x; b[2 * i + 1] = k[good[i].trainIdx].y;
    }
    essential_ransac(*p, a.data(), b.data(), (int)good.size(), res->E, nullptr, &res->n_inliers, &res->iters_run);
    if (res->n_inliers > 0)
        recover_pose(*p, res->E, a.data(), b.data(), (int)good.size(), res->R, res->t, &res->n_pose_good);
    return VIS_OK;
}


// Frame-parallel run of the same per-frame pipeline over a resident stream, for the "all host cores" CPU figure that
// SURVEY section 8(d) asks for beside the single-thread one (OpenCV itself would use TBB inside its calls).
// Phase 1: frames are detected in parallel; phase 2: consecutive pairs are matched + posed in parallel.  Per-frame
// results are identical to calling orc_pipeline_frame frame after frame.  Returns the wall seconds of both phases.
#include <atomic>
#include <chrono>
#include <thread>
extern "C" int orc_pipeline_stream_mt(const vis_params* p, const uint8_t* frames, int n, int w, int h, int stride,
                                      int threads, double* seconds, orc_frame_result* results /* n, may be NULL */) {
    if (!p || !frames || n < 1 || threads < 1 || !seconds) return VIS_E_INVALID;
    std::vector<std::vector<vis_keypoint> > K((size_t)n);
    std::vector<std::vector<uint8_t> > D((size_t)n);
    std::vector<orc_frame_result> R((size_t)n);
    std::atomic<int> next(0), err(0);
    const size_t fbytes = (size_t)stride * h;
    const auto t0 = std::chrono::steady_clock::now();
    auto run = [&](auto fn) {
        next = 0;
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back([&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); });
        for (auto& th : pool) th.join();
    };
    run([&](int i) {
        std::memset(&R[(size_t)i], 0, sizeof(orc_frame_result));
        const uint8_t* img = frames + (size_t)i * fbytes;
        std::vector<uint8_t> lv[5]; uint8_t* lp[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        int32_t hw[5], hh[5];
        orc_half_pyramid_dims(w, h, hw, hh);
        for (int l = 1; l < 5; l++) { lv[l].resize((size_t)hw[l] * hh[l] + 16); lp[l] = lv[l].data(); }
        orc_half_pyramid(img, w, h, stride, lp);
        if (orb_detect_compute(*p, img, w, h, stride, K[(size_t)i], D[(size_t)i])) err = 1;
        R[(size_t)i].n_kp = (int)K[(size_t)i].size();
    });
    run([&](int i) {
        if (i == 0 || K[(size_t)i].empty() || K[(size_t)i - 1].empty()) return;
        const std::vector<vis_keypoint>& pk = K[(size_t)i - 1]; const std::vector<vis_keypoint>& k = K[(size_t)i];
        const std::vector<uint8_t>& pd = D[(size_t)i - 1]; const std::vector<uint8_t>& d = D[(size_t)i];
        orc_frame_result& res = R[(size_t)i];
        std::vector<vis_dmatch> k12(2 * pk.size()), k21(2 * k.size());
        knn2_hamming(pd.data(), (int)pk.size(), d.data(), (int)k.size(), k12.data());
        knn2_hamming(d.data(), (int)k.size(), pd.data(), (int)pk.size(), k21.data());
        std::vector<vis_dmatch> sym, good;
        good_matches(*p, pk.data(), (int)pk.size(), k.data(), (int)k.size(), k12.data(), k21.data(), sym, good);
        res.n_sym = (int)sym.size(); res.n_good = (int)good.size();
        std::vector<float> a(2 * good.size()), b(2 * good.size());
        for (size_t j = 0; j < good.size(); j++) {
            a[2 * j] = pk[good[j].queryIdx].x; a[2 * j + 1] = pk[good[j].queryIdx].y;
            b[2 * j] = k[good[j].trainIdx].x; b[2 * j + 1] = k[good[j].trainIdx].y;
        }
        essential_ransac(*p, a.data(), b.data(), (int)good.size(), res.E, nullptr, &res.n_inliers, &res.iters_run);
        if (res.n_inliers > 0) recover_pose(*p, res.E, a.data(), b.data(), (int)good.size(), res.R, res.t, &res.n_pose_good);
    });
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (results && n > 0) std::memcpy(results, R.data(), (size_t)n * sizeof(orc_frame_result));
    return err ? VIS_E_INVALID : VIS_OK;
}
