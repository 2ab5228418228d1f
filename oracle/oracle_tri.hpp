// ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (see oracle_geom.hpp header).
// CPU restatement of the multi-view triangulation of abignoli/EdgeGraph3D:
//   triangulation.cpp = src/edgegraph3d/utils/geometry/triangulation.cpp
//   gauss_newton.cpp  = src/edgegraph3d/filtering/gauss_newton.cpp
//
// Third-party arithmetic NOT under /root/reference: OpenCV (>=3.1, unpinned; README.md:23,32).
// cv::triangulatePoints, cv::SVD (Jacobi), Mat GEMM, cv::determinant, Mat::inv and
// cv::computeCorrespondEpilines are restated here from the published OpenCV 3.x algorithms
// (modules/calib3d/src/triangulate.cpp, modules/core/src/lapack.cpp JacobiSVDImpl_,
// modules/core/src/matmul.cpp GEMMSingleMul, modules/calib3d/src/fundam.cpp). The
// restatement fixes one evaluation order (documented per function) that the HIP path shares.
// Round 4: compute_projection (vec4 * mat4 of the vendored glm) is PINNED bit for bit against that glm
// (tests/test_glm_pin.py); the DLT, the Jacobi SVD and the Gauss-Newton GEMM orders are OpenCV's and stay unpinned —
// cross-checked numerically against numpy / scipy only (tests/test_dlt_forms.py).
#pragma once
#include <cfloat>
#include <cmath>
#include <vector>

#include "oracle_geom.hpp"

namespace orc {

struct Cameras {
  int n_views;
  const float* P;  // [V][16] row-major cameraMatrix[r][c]
  const double* F; // [V][V][9]
  const uint8_t* F_valid;
};

// geometric_utilities.cpp:824-843 -> cv::computeCorrespondEpilines(points, 1, F, lines):
// l = F * (x, y, 1)^T in double, scaled by 1/sqrt(a^2+b^2), rounded to float.
// TEST HOOK (tools/convention_report.py, orc_set_conventions): how exposed the results are to conventions of the OpenCV
// routines that this restatement could not pin (no OpenCV in the image). 0 = the restatement as is. Bits:
//   1   GEMM sums (J^T J and (H^-1 J^T) r) with TWO interleaved accumulators (even / odd terms), added at the end
//   2   ... with FOUR interleaved accumulators
//   4   Jacobi SVD: gamma = hypot(p, beta) instead of sqrt(p*p + beta*beta)
//   8   Jacobi SVD: rotation pairs visited in the opposite order (i descending, j descending)
//   16  epipolar line normalised by division (a / sqrt(nu)) instead of multiplication by 1 / sqrt(nu)
static unsigned g_conv = 0;
template <class F>
static inline double conv_sum(int count, F term) {  // sum_{k < count} term(k) under the selected GEMM convention
  if (g_conv & 2u) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int k = 0;
    for (; k + 4 <= count; k += 4) {
      s0 += term(k);
      s1 += term(k + 1);
      s2 += term(k + 2);
      s3 += term(k + 3);
    }
    double s = (s0 + s1) + (s2 + s3);
    for (; k < count; k++) s += term(k);
    return s;
  }
  if (g_conv & 1u) {
    double s0 = 0, s1 = 0;
    int k = 0;
    for (; k + 2 <= count; k += 2) {
      s0 += term(k);
      s1 += term(k + 1);
    }
    double s = s0 + s1;
    for (; k < count; k++) s += term(k);
    return s;
  }
  double s = 0;
  for (int k = 0; k < count; k++) s += term(k);
  return s;
}
static inline bool computeCorrespondEpilineSinglePoint(const Cameras& cams, int view_from, int view_to,
                                                       const vec2& p, float epipolar_line[3]) {
  if (!cams.F_valid[(size_t)view_from * cams.n_views + view_to]) return false;  // "F.rows==3 && F.cols==3" fails
  const double* f = cams.F + ((size_t)view_from * cams.n_views + view_to) * 9;
  double t0 = p.x, t1 = p.y;
  double a = (f[0] * t0 + f[1] * t1) + f[2];
  double b = (f[3] * t0 + f[4] * t1) + f[5];
  double c = (f[6] * t0 + f[7] * t1) + f[8];
  double nu = a * a + b * b;
  if ((g_conv & 16u) && nu) {
    const double sq = std::sqrt(nu);
    a /= sq;
    b /= sq;
    c /= sq;
  } else {
    nu = nu ? 1. / std::sqrt(nu) : 1.;
    a *= nu;
    b *= nu;
    c *= nu;
  }
  epipolar_line[0] = (float)a;
  epipolar_line[1] = (float)b;
  epipolar_line[2] = (float)c;
  return true;
}

// geometric_utilities.cpp:973-977 — vec4 * mat4 of glm 0.9.6 (type_mat4x4.inl:640-651):
// u_r = ((P[r][0]*X + P[r][1]*Y) + P[r][2]*Z) + P[r][3]*1, all float (Q5, Q6)
static inline vec2 compute_projection(const float* P, const vec3& X) {
  float u0 = ((P[0] * X.x + P[1] * X.y) + P[2] * X.z) + P[3] * 1.0f;
  float u1 = ((P[4] * X.x + P[5] * X.y) + P[6] * X.z) + P[7] * 1.0f;
  float u2 = ((P[8] * X.x + P[9] * X.y) + P[10] * X.z) + P[11] * 1.0f;
  return vec2(u0 / u2, u1 / u2);
}

// Which linear system cv::triangulatePoints builds depends on the OpenCV release, and the
// reference does not pin one ("OpenCV >= 3.1, tested 3.1", README.md:23,32):
//   rows per view = 3 : OpenCV 2.4 .. 3.1 (legacy cvTriangulatePoints, `double matrA_dat[24]`): a 6x4
//       system, rows x*P2-P0, y*P2-P1, x*P1-y*P0 per view, cvSVD on the 6x4 matrix;
//   rows per view = 2 : the later rewrite (Matx<double,4,4>): a 4x4 system without the third row.
// Both are restated (from knowledge of the sources — none are in this container); the switch is
// orc_set_dlt_rows(), the product's compile-time EG3D_DLT_ROWS. Only the Gauss-Newton START changes.
static int g_dlt_rows = 3;  // the form of the OpenCV release the reference names (3.1); see eg3d_dev_tri.h

// One-sided (Hestenes) Jacobi SVD of an m x 4 double matrix (m = 4 or 6), restating OpenCV's
// JacobiSVDImpl_<double> as reached from cvSVD / SVD::compute inside cvTriangulatePoints: for
// m >= n the routine works on At = A transposed (n = 4 rows of length m), W[i] = |row i|^2, and
// rotates row pairs (i < j ascending) until no pair changes (at most max(m,30) = 30 sweeps).
// Returns the right singular vector of the smallest singular value = last row of Vt after the
// descending sort. hypot() is replaced by sqrt(p*p+beta*beta) (documented deviation: libm hypot
// is not available to the device code).
static inline void jacobi_svd_last_v(const double At_in[4][6], int m, double out[4]) {
  const int n = 4;
  double At[4][6], Vt[4][4], W[4];
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < m; k++) At[i][k] = At_in[i][k];
  const double eps = DBL_EPSILON * 10;
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) {
      double t = At[i][k];
      sd += t * t;
    }
    W[i] = sd;
    for (int k = 0; k < n; k++) Vt[i][k] = 0;
    Vt[i][i] = 1;
  }
  const int max_iter = 30;  // std::max(m, 30)
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    // the pairs (i < j) of n = 4 rows in ascending order: i outer, j inner (test hook bit 8: the same pairs, last first)
    static const int PAIR_I[6] = {0, 0, 0, 1, 1, 2}, PAIR_J[6] = {1, 2, 3, 2, 3, 3};
    for (int pq = 0; pq < 6; pq++) {
      {
        const int pr = (g_conv & 8u) ? 5 - pq : pq;
        const int i = PAIR_I[pr], j = PAIR_J[pr];
        double* Ai = At[i];
        double* Aj = At[j];
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (std::fabs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = (g_conv & 4u) ? std::hypot(p, beta) : std::sqrt(p * p + beta * beta);
        double c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = std::sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          double t0 = c * Ai[k] + s * Aj[k];
          double t1 = c * Aj[k] - s * Ai[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        double* Vi = Vt[i];
        double* Vj = Vt[j];
        for (int k = 0; k < n; k++) {
          double t0 = c * Vi[k] + s * Vj[k];
          double t1 = c * Vj[k] - s * Vi[k];
          Vi[k] = t0;
          Vj[k] = t1;
        }
      }
    }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) {
      double t = At[i][k];
      sd += t * t;
    }
    W[i] = std::sqrt(sd);
  }
  // selection sort, descending, swapping the Vt rows along (lapack.cpp JacobiSVDImpl_)
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < n; k++) std::swap(Vt[i][k], Vt[j][k]);
    }
  }
  for (int k = 0; k < 4; k++) out[k] = Vt[3][k];
}

// cv::triangulatePoints for one point pair (triangulate.cpp cvTriangulatePoints): per view j the
// rows x_j*P_j(2,:) - P_j(0,:), y_j*P_j(2,:) - P_j(1,:) [and x_j*P_j(1,:) - y_j*P_j(0,:) in the
// three-row form] in double, solution = V[:,3]; the 4x1 output Mat is CV_32F (same type as the
// input points), so each homogeneous component is rounded to float before the caller divides by w
// in float (triangulation.cpp:216-224).
static inline void dlt2_init(const float* P1, const vec2& p1, const float* P2, const vec2& p2, double X0[3]) {
  double At[4][6];  // At[k][row] = A[row][k]
  const float* Ps[2] = {P1, P2};
  const vec2 pts[2] = {p1, p2};
  const int R = g_dlt_rows;
  for (int j = 0; j < 2; j++) {
    double x = pts[j].x, y = pts[j].y;
    for (int k = 0; k < 4; k++) {
      At[k][j * R + 0] = x * (double)Ps[j][8 + k] - (double)Ps[j][0 + k];
      At[k][j * R + 1] = y * (double)Ps[j][8 + k] - (double)Ps[j][4 + k];
      if (R == 3) At[k][j * R + 2] = x * (double)Ps[j][4 + k] - y * (double)Ps[j][0 + k];
    }
  }
  double v[4];
  jacobi_svd_last_v(At, 2 * R, v);
  float h0 = (float)v[0], h1 = (float)v[1], h2 = (float)v[2], h3 = (float)v[3];
  X0[0] = (double)(h0 / h3);
  X0[1] = (double)(h1 / h3);
  X0[2] = (double)(h2 / h3);
}

struct GNObs {
  const float* P;  // 4x4 float camera matrix, widened to double on use (triangulation.cpp:230-233)
  float x, y;      // cv::Point2f
};

// em_GaussNewton + em_point2D3DJacobian, triangulation.cpp:105-176 and :53-103, FP64.
// GEMM orders restated from OpenCV matmul.cpp: 4x4*4x1 products sum left to right;
// H = J^T J sums over the 2n rows in order from 0; (H^-1 * J^T) is formed first, then
// multiplied by r summing over the 2n columns in order.
static inline int em_GaussNewton(const std::vector<GNObs>& obs, const double init[3], double out[3]) {
  const int n = (int)obs.size();
  // scratch of the solve, per thread and reused (the values are overwritten before they are read: same arithmetic; a
  // std::vector pair per call was ~50 M malloc / free per C3' step, serialising the threads of an all-core run)
  static thread_local std::vector<double> r_tl, J_tl;
  if (r_tl.size() < (size_t)(2 * n)) r_tl.resize(2 * n);
  if (J_tl.size() < (size_t)(6 * n)) J_tl.resize(6 * n);
  double* const r = r_tl.data();
  double* const J = J_tl.data();
  double X[3] = {init[0], init[1], init[2]};
  double last_mse = 0;
  for (int it = 0; it < 30; it++) {
    double mse = 0;
    for (int m = 0; m < n; m++) {
      const float* P = obs[m].P;
      double h0 = (((double)P[0] * X[0] + (double)P[1] * X[1]) + (double)P[2] * X[2]) + (double)P[3] * 1.0;
      double h1 = (((double)P[4] * X[0] + (double)P[5] * X[1]) + (double)P[6] * X[2]) + (double)P[7] * 1.0;
      double h2 = (((double)P[8] * X[0] + (double)P[9] * X[1]) + (double)P[10] * X[2]) + (double)P[11] * 1.0;
      r[2 * m] = (double)obs[m].x - h0 / h2;
      mse += r[2 * m] * r[2 * m];
      r[2 * m + 1] = (double)obs[m].y - h1 / h2;
      mse += r[2 * m + 1] * r[2 * m + 1];
    }
    if (std::fabs(mse / (n * 2) - last_mse) < 0.0000005) break;
    last_mse = mse / (n * 2);
    // em_point2D3DJacobian
    for (int m = 0; m < n; m++) {
      const float* P = obs[m].P;
      double xH = (((double)P[0] * X[0] + (double)P[1] * X[1]) + (double)P[2] * X[2]) + (double)P[3] * 1.0;
      double yH = (((double)P[4] * X[0] + (double)P[5] * X[1]) + (double)P[6] * X[2]) + (double)P[7] * 1.0;
      double zH = (((double)P[8] * X[0] + (double)P[9] * X[1]) + (double)P[10] * X[2]) + (double)P[11] * 1.0;
      double p00 = P[0], p01 = P[1], p02 = P[2], p10 = P[4], p11 = P[5], p12 = P[6], p20 = P[8], p21 = P[9],
             p22 = P[10];
      double zz = zH * zH;
      J[(2 * m) * 3 + 0] = (p00 * zH - p20 * xH) / zz;
      J[(2 * m + 1) * 3 + 0] = (p10 * zH - p20 * yH) / zz;
      J[(2 * m) * 3 + 1] = (p01 * zH - p21 * xH) / zz;
      J[(2 * m + 1) * 3 + 1] = (p11 * zH - p21 * yH) / zz;
      J[(2 * m) * 3 + 2] = (p02 * zH - p22 * xH) / zz;
      J[(2 * m + 1) * 3 + 2] = (p12 * zH - p22 * yH) / zz;
    }
    double H[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        H[i][j] = g_conv & 3u ? conv_sum(2 * n, [&](int k) { return J[k * 3 + i] * J[k * 3 + j]; }) : [&]() {
          double s = 0;
          for (int k = 0; k < 2 * n; k++) s += J[k * 3 + i] * J[k * 3 + j];
          return s;
        }();
      }
    // cv::determinant 3x3 (det3 macro)
    double d = H[0][0] * (H[1][1] * H[2][2] - H[1][2] * H[2][1]) - H[0][1] * (H[1][0] * H[2][2] - H[1][2] * H[2][0]) +
               H[0][2] * (H[1][0] * H[2][1] - H[1][1] * H[2][0]);
    if (d < 0.00001) return -1;
    // Mat::inv() DECOMP_LU, n==3 closed form (lapack.cpp cv::invert)
    double Hi[3][3];
    {
      double id = 1. / d;
      Hi[0][0] = (H[1][1] * H[2][2] - H[1][2] * H[2][1]) * id;
      Hi[0][1] = (H[0][2] * H[2][1] - H[0][1] * H[2][2]) * id;
      Hi[0][2] = (H[0][1] * H[1][2] - H[0][2] * H[1][1]) * id;
      Hi[1][0] = (H[1][2] * H[2][0] - H[1][0] * H[2][2]) * id;
      Hi[1][1] = (H[0][0] * H[2][2] - H[0][2] * H[2][0]) * id;
      Hi[1][2] = (H[0][2] * H[1][0] - H[0][0] * H[1][2]) * id;
      Hi[2][0] = (H[1][0] * H[2][1] - H[1][1] * H[2][0]) * id;
      Hi[2][1] = (H[0][1] * H[2][0] - H[0][0] * H[2][1]) * id;
      Hi[2][2] = (H[0][0] * H[1][1] - H[0][1] * H[1][0]) * id;
    }
    // X += (H^-1 * J^T) * r
    for (int i = 0; i < 3; i++) {
      auto term = [&](int k) {
        double mik = (Hi[i][0] * J[k * 3 + 0] + Hi[i][1] * J[k * 3 + 1]) + Hi[i][2] * J[k * 3 + 2];
        return mik * r[k];
      };
      double s = 0;
      if (g_conv & 3u)
        s = conv_sum(2 * n, term);
      else
        for (int k = 0; k < 2 * n; k++) s += term(k);
      X[i] += s;
    }
  }
  if (last_mse < 9) {
    out[0] = X[0];
    out[1] = X[1];
    out[2] = X[2];
    return 1;
  }
  return -1;
}

// edge_graph_3d_utilities.hpp:69-92 — Q1: max_index is always the last index.
static inline std::pair<int, int> get_min_max(const std::vector<int>& vals) {
  int min_index = 0;
  int mn = vals[0];
  for (int i = 0; i < (int)vals.size(); i++)
    if (vals[i] < mn) {
      mn = vals[i];
      min_index = i;
    }
  return std::make_pair(min_index, (int)vals.size() - 1);
}

struct TriStats {
  uint64_t n_tri = 0, n_add = 0, n_degenerate_dlt = 0, n_combos = 0;
};

// em_estimate3Dpositions, triangulation.cpp:178-250 / :252-323 (both overloads do the same
// arithmetic). coords/ids in list order.
static inline void em_estimate3Dpositions(const Cameras& cams, const std::vector<vec2>& coords,
                                          const std::vector<int>& ids, vec3& triangulated_point, bool& valid,
                                          TriStats* st) {
  const std::pair<int, int> mm = get_min_max(ids);
  int firstCamIdx = ids[mm.first];
  int lastCamIdx = ids[mm.second];
  if (st) {
    st->n_tri++;
    if (firstCamIdx == lastCamIdx) st->n_degenerate_dlt++;  // Q11
  }
  double init[3], opt[3];
  dlt2_init(cams.P + (size_t)firstCamIdx * 16, coords[mm.first], cams.P + (size_t)lastCamIdx * 16,
            coords[mm.second], init);
  static thread_local std::vector<GNObs> obs;
  obs.resize(ids.size());
  for (size_t i = 0; i < ids.size(); i++) {
    obs[i].P = cams.P + (size_t)ids[i] * 16;
    obs[i].x = coords[i].x;
    obs[i].y = coords[i].y;
  }
  int res = em_GaussNewton(obs, init, opt);
  if (res != -1) {
    triangulated_point = vec3((float)opt[0], (float)opt[1], (float)opt[2]);
    valid = true;
  } else
    valid = false;
}

// em_add_new_observation_to_3Dpositions, triangulation.cpp:347-405 / :408-466: GN from the
// stored (float) X with all current observations plus the new one — no DLT.
static inline void em_add_new_observation_to_3Dpositions(const Cameras& cams, const vec3& cur_X,
                                                         const std::vector<vec2>& cur_coords,
                                                         const std::vector<int>& cur_ids, const vec2 new_coords,
                                                         const int new_view, vec3& triangulated_point, bool& valid,
                                                         TriStats* st) {
  if (st) st->n_add++;
  double init[3] = {(double)cur_X.x, (double)cur_X.y, (double)cur_X.z}, opt[3];
  static thread_local std::vector<GNObs> obs;
  obs.resize(cur_ids.size() + 1);
  for (size_t i = 0; i < cur_ids.size(); i++) {
    obs[i].P = cams.P + (size_t)cur_ids[i] * 16;
    obs[i].x = cur_coords[i].x;
    obs[i].y = cur_coords[i].y;
  }
  obs[cur_ids.size()].P = cams.P + (size_t)new_view * 16;
  obs[cur_ids.size()].x = new_coords.x;
  obs[cur_ids.size()].y = new_coords.y;
  int res = em_GaussNewton(obs, init, opt);
  if (res != -1) {
    triangulated_point = vec3((float)opt[0], (float)opt[1], (float)opt[2]);
    valid = true;
  } else
    valid = false;
}

// compute_3d_point_coords, triangulation.cpp:468-481
static inline void compute_3d_point_coords(const Cameras& cams, const std::vector<vec2>& coords,
                                           const std::vector<int>& ids, vec3& new_point, bool& valid, TriStats* st) {
  valid = false;
  if (coords.size() >= 2) em_estimate3Dpositions(cams, coords, ids, new_point, valid, st);
}

// compute_3d_point_coords_combinations, triangulation.cpp:1105-1158. Only `selected`,
// `new_point` and `valid` are consumed by the caller (plg_matching.cpp:733-752); the
// re-ordering at :1155-1157 touches vectors the caller discards.
static inline void compute_3d_point_coords_combinations(const Cameras& cams, const std::vector<vec2>& all_coords,
                                                        const std::vector<int>& all_ids, const int min_combinations,
                                                        std::vector<bool>& selected, vec3& new_point, bool& valid,
                                                        TriStats* st) {
  valid = false;
  if (st) st->n_combos++;
  std::vector<vec2> sel_coords;
  std::vector<int> sel_ids;
  selected.resize(all_ids.size());
  std::fill(selected.begin() + min_combinations, selected.end(), false);
  std::fill(selected.begin(), selected.begin() + min_combinations, true);
  do {
    sel_coords.clear();
    sel_ids.clear();
    for (size_t i = 0; i < all_ids.size(); ++i)
      if (selected[i]) {
        sel_coords.push_back(all_coords[i]);
        sel_ids.push_back(all_ids[i]);
      }
    compute_3d_point_coords(cams, sel_coords, sel_ids, new_point, valid, st);
  } while (!valid && std::prev_permutation(selected.begin(), selected.end()));
  if (!valid) return;
  vec3 new_3d;
  for (size_t i = 0; i < all_ids.size(); ++i) {
    if (!selected[i]) {
      em_add_new_observation_to_3Dpositions(cams, new_point, sel_coords, sel_ids, all_coords[i], all_ids[i], new_3d,
                                            valid, st);
      if (valid) {
        selected[i] = true;
        new_point = new_3d;
        sel_coords.push_back(all_coords[i]);
        sel_ids.push_back(all_ids[i]);
      }
    }
  }
  valid = true;
}

// ------------------------------------------------------------ config 5 (FP32) ----
// GaussNewton + point2D3DJacobian, gauss_newton.cpp:83-134 and :26-75, CV_32F Mats.
// OpenCV float GEMM: the <=4 fast path (4x4*4x1, and 3x3*3x2n when 2n<=16) is float
// arithmetic; the generic path (J^T J, and everything*r) accumulates in double and rounds
// each result to float. determinant/invert of a 3x3 float Mat evaluate in double.
static inline int GaussNewton_f32(const std::vector<GNObs>& obs, const float init[3], float out[3],
                                  const float gn_max_mse, bool legacy_abs) {
  const int n = (int)obs.size();
  std::vector<float> r(2 * n), J(6 * n);
  float X[3] = {init[0], init[1], init[2]};
  float last_mse = 0;
  for (int it = 0; it < 30; it++) {
    float mse = 0;
    for (int m = 0; m < n; m++) {
      const float* P = obs[m].P;
      float h0 = ((P[0] * X[0] + P[1] * X[1]) + P[2] * X[2]) + P[3] * 1.0f;
      float h1 = ((P[4] * X[0] + P[5] * X[1]) + P[6] * X[2]) + P[7] * 1.0f;
      float h2 = ((P[8] * X[0] + P[9] * X[1]) + P[10] * X[2]) + P[11] * 1.0f;
      r[2 * m] = obs[m].x - h0 / h2;
      mse += r[2 * m] * r[2 * m];
      r[2 * m + 1] = obs[m].y - h1 / h2;
      mse += r[2 * m + 1] * r[2 * m + 1];
    }
    float diff = mse / (n * 2) - last_mse;
    bool conv;
    if (legacy_abs)
      // Q9: ::abs(int) of the truncated difference; 0 exactly when -1 < diff < 1. For NaN / out-of-range
      // differences (0 observations: 0/0) the reference's float -> int conversion is undefined; x86's cvttss2si
      // gives INT_MIN, abs() of which is not 0: "not converged". Stated as a predicate so that neither the
      // compiler's treatment of abs(INT_MIN) nor another target's conversion can change it.
      conv = diff > -1.0f && diff < 1.0f;
    else
      conv = (double)std::fabs(diff) < 0.0000000005;
    if (conv) break;
    last_mse = mse / (n * 2);
    for (int m = 0; m < n; m++) {
      const float* P = obs[m].P;
      float xH = ((P[0] * X[0] + P[1] * X[1]) + P[2] * X[2]) + P[3] * 1.0f;
      float yH = ((P[4] * X[0] + P[5] * X[1]) + P[6] * X[2]) + P[7] * 1.0f;
      float zH = ((P[8] * X[0] + P[9] * X[1]) + P[10] * X[2]) + P[11] * 1.0f;
      float zz = zH * zH;
      J[(2 * m) * 3 + 0] = (P[0] * zH - P[8] * xH) / zz;
      J[(2 * m + 1) * 3 + 0] = (P[4] * zH - P[8] * yH) / zz;
      J[(2 * m) * 3 + 1] = (P[1] * zH - P[9] * xH) / zz;
      J[(2 * m + 1) * 3 + 1] = (P[5] * zH - P[9] * yH) / zz;
      J[(2 * m) * 3 + 2] = (P[2] * zH - P[10] * xH) / zz;
      J[(2 * m + 1) * 3 + 2] = (P[6] * zH - P[10] * yH) / zz;
    }
    float H[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 2 * n; k++) s += (double)J[k * 3 + i] * (double)J[k * 3 + j];
        H[i][j] = (float)s;
      }
    double dd = H[0][0] * ((double)H[1][1] * H[2][2] - (double)H[1][2] * H[2][1]) -
                H[0][1] * ((double)H[1][0] * H[2][2] - (double)H[1][2] * H[2][0]) +
                H[0][2] * ((double)H[1][0] * H[2][1] - (double)H[1][1] * H[2][0]);
    float d = (float)dd;
    if ((double)d < 0.0000000001) return -1;
    float Hi[3][3];
    if (dd != 0.) {
      double id = 1. / dd;
      Hi[0][0] = (float)(((double)H[1][1] * H[2][2] - (double)H[1][2] * H[2][1]) * id);
      Hi[0][1] = (float)(((double)H[0][2] * H[2][1] - (double)H[0][1] * H[2][2]) * id);
      Hi[0][2] = (float)(((double)H[0][1] * H[1][2] - (double)H[0][2] * H[1][1]) * id);
      Hi[1][0] = (float)(((double)H[1][2] * H[2][0] - (double)H[1][0] * H[2][2]) * id);
      Hi[1][1] = (float)(((double)H[0][0] * H[2][2] - (double)H[0][2] * H[2][0]) * id);
      Hi[1][2] = (float)(((double)H[0][2] * H[1][0] - (double)H[0][0] * H[1][2]) * id);
      Hi[2][0] = (float)(((double)H[1][0] * H[2][1] - (double)H[1][1] * H[2][0]) * id);
      Hi[2][1] = (float)(((double)H[0][1] * H[2][0] - (double)H[0][0] * H[2][1]) * id);
      Hi[2][2] = (float)(((double)H[0][0] * H[1][1] - (double)H[0][1] * H[1][0]) * id);
    } else {
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Hi[i][j] = 0;
    }
    for (int i = 0; i < 3; i++) {
      double s = 0;
      for (int k = 0; k < 2 * n; k++) {
        // H.inv()*J.t() is gemm(Hinv, J, GEMM_2_T): flags != 0, so the generic path with
        // double accumulators runs, each element rounded to float
        double a = ((double)Hi[i][0] * (double)J[k * 3 + 0] + (double)Hi[i][1] * (double)J[k * 3 + 1]) +
                   (double)Hi[i][2] * (double)J[k * 3 + 2];
        float mik = (float)a;
        s += (double)mik * (double)r[k];
      }
      X[i] += (float)s;
    }
  }
  if (last_mse < gn_max_mse) {
    out[0] = X[0];
    out[1] = X[1];
    out[2] = X[2];
    return 1;
  }
  return -1;
}

}  // namespace orc
