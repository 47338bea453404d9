// oracle/oracle_internal.h -- CPU ORACLE internals (test infrastructure only).
#ifndef ORACLE_INTERNAL_H_
#define ORACLE_INTERNAL_H_
#include "vis_oracle.h"
#include <vector>

namespace orc {
struct LevelGeom { int w, h, quota; float scale; };
struct RawKp { int x, y; float response; };

void sincos_det(double x, double* s, double* c);
float fast_atan2(float y, float x);
void compute_umax(int halfPatch, std::vector<int>& umax);
void gaussian_kernel7_q8(int k[7]);
void level_geometry(const vis_params& p, int w, int h, std::vector<LevelGeom>& g);
int orb_detect_compute(const vis_params& p, const uint8_t* img, int w, int h, int stride,
                       std::vector<vis_keypoint>& kps, std::vector<uint8_t>& desc);
void knn2_hamming(const uint8_t* dq, int nq, const uint8_t* dt, int nt, vis_dmatch* out);
int good_matches(const vis_params& p, const vis_keypoint* kps1, int n1, const vis_keypoint* kps2, int n2,
                 const vis_dmatch* knn12, const vis_dmatch* knn21,
                 std::vector<vis_dmatch>& sym, std::vector<vis_dmatch>& good);
}  // namespace orc
#endif
