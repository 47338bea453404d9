// oracle/match.cpp -- CPU ORACLE (test infrastructure only; see vis_oracle.h header).
//
// Restates Matcher::computeMatches (BFMatcher(NORM_HAMMING)::knnMatch k=2, both directions,
// /root/reference/src/Matcher.cpp:83-94; GPU twin src/MatcherGPU.cpp:44-66) and the reference's
// own post-filters: nnFilter (:148-169), computeSymMatches (:96-144), sortMatches (:329-352),
// bestMatchesFilter (:171-244), getGoodMatches (:295-303).  Where the reference has undefined
// behaviour the resolution specified in SURVEY.md section 8(a) is implemented and marked "SPEC".
#include "vis_oracle.h"
#include "oracle_internal.h"
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

namespace orc {

static inline int hamming256(const uint8_t* a, const uint8_t* b) {
    uint64_t x[4], y[4];
    std::memcpy(x, a, 32); std::memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

// Appendix A.2: cv::batchDistance k-NN update: scan train rows ascending, K=2 sorted slots
// initialised (INT_MAX,-1); insert when d < slot[K-1], shifting while slot[k] > d (strict),
// so equal distances keep the lower train index first.  out: nq x 2.
void knn2_hamming(const uint8_t* dq, int nq, const uint8_t* dt, int nt, vis_dmatch* out) {
    for (int q = 0; q < nq; q++) {
        int bd[2] = {INT_MAX, INT_MAX}, bi[2] = {-1, -1};
        for (int t = 0; t < nt; t++) {
            int d = hamming256(dq + 32 * (size_t)q, dt + 32 * (size_t)t);
            if (d < bd[1]) {
                if (d < bd[0]) { bd[1] = bd[0]; bi[1] = bi[0]; bd[0] = d; bi[0] = t; }
                else { bd[1] = d; bi[1] = t; }
            }
        }
        for (int k = 0; k < 2; k++) {
            vis_dmatch& m = out[2 * (size_t)q + k];
            m.queryIdx = q; m.trainIdx = bi[k];
            m.imgIdx = bi[k] >= 0 ? 0 : -1;
            m.distance = bi[k] >= 0 ? (float)bd[k] : FLT_MAX;
        }
    }
}

// nnFilter (src/Matcher.cpp:148-169): survives iff 2 neighbours and !(d0 > ratio*d1), the
// product taken in double with ratio = (double)0.8f (src/Matcher.cpp:103).
static inline bool ratio_survives(const vis_dmatch* two, double nn_ratio) {
    if (two[0].trainIdx < 0 || two[1].trainIdx < 0) return false;
    return !((double)two[0].distance > nn_ratio * (double)two[1].distance);
}

int good_matches(const vis_params& p, const vis_keypoint* kps1, int n1, const vis_keypoint* kps2, int n2,
                 const vis_dmatch* knn12, const vis_dmatch* knn21,
                 std::vector<vis_dmatch>& sym, std::vector<vis_dmatch>& good) {
    sym.clear(); good.clear();
    const double nn_ratio = (double)p.ratio;
    // ---- computeSymMatches (src/Matcher.cpp:96-144).  SPEC: the inner guard
    // re-tests aux1 (:122) so direction 2 is effectively NOT ratio filtered (its cleared vectors
    // are read through stale storage).  REFERENCE_EFFECTIVE = mutual best + ratio on direction 1;
    // INTENDED = ratio on both.  Output ordered by queryIdx ascending; DMatch(q,t,dist) => imgIdx -1.
    for (int q = 0; q < n1; q++) {
        const vis_dmatch* a = knn12 + 2 * (size_t)q;
        if (!ratio_survives(a, nn_ratio)) continue;
        int t = a[0].trainIdx;
        if (t < 0 || t >= n2) continue;
        const vis_dmatch* b = knn21 + 2 * (size_t)t;
        if (b[0].trainIdx != q) continue;
        if (p.sym_mode == VIS_SYM_INTENDED && !ratio_survives(b, nn_ratio)) continue;
        vis_dmatch m; m.queryIdx = q; m.trainIdx = t; m.imgIdx = -1; m.distance = a[0].distance;
        sym.push_back(m);
    }
    // ---- sortMatches (src/Matcher.cpp:329-352): argsort by keypoints_1[queryIdx].pt.y ascending.
    // SPEC (a10): cv::sortIdx is unstable -> stable sort, ties keep queryIdx order.
    std::vector<vis_dmatch> sorted = sym;
    std::stable_sort(sorted.begin(), sorted.end(), [&](const vis_dmatch& A, const vis_dmatch& B) {
        return kps1[A.queryIdx].y < kps1[B.queryIdx].y;
    });
    // ---- bestMatchesFilter (src/Matcher.cpp:171-244), literal control flow.
    // SPEC (a11): empty input returns 0; column index clamped to root-1.
    if (sorted.empty()) return 0;
    const int n_features = p.n_cells;
    float winWSize = (float)(p.w_size / std::floor(std::sqrt((double)n_features)));
    float winHSize = (float)(p.h_size / std::floor(std::sqrt((double)n_features)));
    int root_n = (int)std::floor(std::sqrt((double)n_features));
    if (root_n < 1) return 0;
    std::vector<vis_dmatch> cells(root_n);
    for (auto& c : cells) { c.queryIdx = -1; c.trainIdx = -1; c.imgIdx = -1; c.distance = 100000.0f; }
    size_t it = 0;
    float h_final = winHSize;
    for (int j = 0; j < root_n; j++) {
        while (kps1[sorted[it].queryIdx].y <= h_final) {
            float w_final = winWSize;
            int i = 0;
            while (kps1[sorted[it].queryIdx].x > w_final) {
                w_final = w_final + winWSize;
                i++;
                if (i >= root_n - 1) { i = root_n - 1; break; }      // SPEC clamp
            }
            if (sorted[it].distance < cells[i].distance) cells[i] = sorted[it];
            ++it;
            if (it == sorted.size()) break;
        }
        for (auto& c : cells) if (c.distance != 100000.0f) good.push_back(c);   // pushBackVectorMatches
        for (auto& c : cells) c.distance = 100000.0f;                            // resetVectorMatches
        h_final = h_final + winHSize;
        if (it == sorted.size()) break;
    }
    (void)kps2; (void)n2;
    return (int)good.size();
}

}  // namespace orc

using namespace orc;

extern "C" int orc_knn2_hamming(const uint8_t* d1, int n1, const uint8_t* d2, int n2,
                                vis_dmatch* out12, vis_dmatch* out21) {
    if (n1 < 0 || n2 < 0 || (n1 && !d1) || (n2 && !d2)) return VIS_E_INVALID;
    if (out12) knn2_hamming(d1, n1, d2, n2, out12);
    if (out21) knn2_hamming(d2, n2, d1, n1, out21);
    return VIS_OK;
}

extern "C" int orc_good_matches(const vis_params* p, const vis_keypoint* kps1, int n1,
                                const vis_keypoint* kps2, int n2,
                                const vis_dmatch* knn12, const vis_dmatch* knn21,
                                vis_dmatch* good, int cap, int* n_good,
                                vis_dmatch* sym_out, int sym_cap, int* n_sym) {
    if (!p || n1 < 0 || n2 < 0) return VIS_E_INVALID;
    std::vector<vis_dmatch> sym, g;
    good_matches(*p, kps1, n1, kps2, n2, knn12, knn21, sym, g);
    if (n_good) *n_good = (int)g.size();
    if (n_sym) *n_sym = (int)sym.size();
    if (good) { if ((int)g.size() > cap) return VIS_E_CAPACITY; if (!g.empty()) std::memcpy(good, g.data(), g.size() * sizeof(vis_dmatch)); }
    if (sym_out) { if ((int)sym.size() > sym_cap) return VIS_E_CAPACITY; if (!sym.empty()) std::memcpy(sym_out, sym.data(), sym.size() * sizeof(vis_dmatch)); }
    return VIS_OK;
}
