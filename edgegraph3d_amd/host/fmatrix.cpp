// Row N4: fundamental matrices from the point tracks of the SfM input
// (generate_all_fundamental_matrices -> generate_all_fundamental_matrices_from_Points ->
// findFundamentalMatrixFromPoints, geometric_utilities.cpp:754-820; helpers
// edge_graph_3d_utilities.cpp:345-352,369-393).
//
// What is reproduced exactly: per ORDERED pair (i, j), i != j, the common points are the ids seen
// from both views in ascending id order; a point's 2-D position on a view is the LAST listed
// observation with that view id; fewer than 10 common points leave the pair without a
// matrix (the reference stores a 1x1 Mat, and every epipolar line of that pair then fails,
// geometric_utilities.cpp:826,840).
// What cannot be reproduced: the matrix itself. The reference calls cv::findFundamentalMat(...,
// FM_LMEDS), a randomised estimator driven by OpenCV's RNG over 7-point minimal samples; OpenCV
// is not available here. This file has the build's own least-median-of-squares estimator with the
// same structure (random minimal samples, median of the larger squared point-to-epipolar-line
// distance, sigma = 2.5 * 1.4826 * (1 + 5 / (n - m)) * sqrt(median) inlier rule) on normalised
// 8-point samples, a deterministic SplitMix64 stream per pair, and a final normalised 8-point fit
// on the inliers. Convention: l_j = F[i][j] * x_i, as eg3d_scene.F.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "eg3d_host.h"

namespace {

constexpr int kMinCommon = 10;  // MIN_CORRESPONDENCES_AMOUNT, geometric_utilities.cpp:752
constexpr int kSample = 8;
constexpr int kIterations = 300;  // log(1-0.99) / log(1 - 0.55^7), the count OpenCV's LMedS settles on

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
};

// Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 9): eigenvalues on the
// diagonal of a, eigenvectors in the columns of v.
template <int N>
void jacobi_eigen(double a[N][N], double v[N][N]) {
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) v[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int p = 0; p < N; p++) {
      diag += a[p][p] * a[p][p];
      for (int q = p + 1; q < N; q++) off += a[p][q] * a[p][q];
    }
    if (off <= 1e-30 * diag) break;
    for (int p = 0; p < N; p++)
      for (int q = p + 1; q < N; q++) {
        if (std::fabs(a[p][q]) < 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; k++) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; k++) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; k++) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
}

struct Pt {
  double x, y;
};

// Hartley normalisation: centroid to the origin, mean distance sqrt(2). T = [[s,0,-s cx],[0,s,-s cy],[0,0,1]].
void normalisation(const Pt* p, const int* idx, int n, double& s, double& cx, double& cy) {
  cx = cy = 0;
  for (int k = 0; k < n; k++) {
    cx += p[idx[k]].x;
    cy += p[idx[k]].y;
  }
  cx /= n;
  cy /= n;
  double d = 0;
  for (int k = 0; k < n; k++) d += std::hypot(p[idx[k]].x - cx, p[idx[k]].y - cy);
  d /= n;
  s = d > 1e-12 ? std::sqrt(2.0) / d : 1.0;
}

// Normalised 8-point fit on the points idx[0..n) (n >= 8): x2' F x1 = 0, rank 2 enforced. False on a
// degenerate sample.
bool eight_point(const Pt* p1, const Pt* p2, const int* idx, int n, double F[9]) {
  double s1, cx1, cy1, s2, cx2, cy2;
  normalisation(p1, idx, n, s1, cx1, cy1);
  normalisation(p2, idx, n, s2, cx2, cy2);
  double A[9][9];
  memset(A, 0, sizeof A);
  for (int k = 0; k < n; k++) {
    const double x1 = (p1[idx[k]].x - cx1) * s1, y1 = (p1[idx[k]].y - cy1) * s1;
    const double x2 = (p2[idx[k]].x - cx2) * s2, y2 = (p2[idx[k]].y - cy2) * s2;
    const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
    for (int a = 0; a < 9; a++)
      for (int b = 0; b < 9; b++) A[a][b] += r[a] * r[b];
  }
  double V[9][9];
  jacobi_eigen<9>(A, V);
  int lo = 0;
  for (int a = 1; a < 9; a++)
    if (A[a][a] < A[lo][lo]) lo = a;
  double Fn[3][3];
  for (int a = 0; a < 9; a++) Fn[a / 3][a % 3] = V[a][lo];
  // rank 2: remove the component along the right singular vector of the smallest singular value
  double G[3][3], W[3][3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double t = 0;
      for (int k = 0; k < 3; k++) t += Fn[k][a] * Fn[k][b];
      G[a][b] = t;
    }
  jacobi_eigen<3>(G, W);
  int l3 = 0;
  for (int a = 1; a < 3; a++)
    if (G[a][a] < G[l3][l3]) l3 = a;
  double Fv[3];
  for (int a = 0; a < 3; a++) Fv[a] = Fn[a][0] * W[0][l3] + Fn[a][1] * W[1][l3] + Fn[a][2] * W[2][l3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) Fn[a][b] -= Fv[a] * W[b][l3];
  // denormalise: F = T2' Fn T1
  const double T1[3][3] = {{s1, 0, -s1 * cx1}, {0, s1, -s1 * cy1}, {0, 0, 1}};
  const double T2[3][3] = {{s2, 0, -s2 * cx2}, {0, s2, -s2 * cy2}, {0, 0, 1}};
  double M[3][3];
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double t = 0;
      for (int k = 0; k < 3; k++) t += Fn[a][k] * T1[k][b];
      M[a][b] = t;
    }
  double nrm = 0;
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      double t = 0;
      for (int k = 0; k < 3; k++) t += T2[k][a] * M[k][b];
      F[3 * a + b] = t;
      nrm += t * t;
    }
  if (!(nrm > 0) || !std::isfinite(nrm)) return false;
  // scale as OpenCV reports it (F33 = 1) when that entry is not tiny, unit Frobenius norm otherwise
  const double sc = std::fabs(F[8]) > 1e-12 * std::sqrt(nrm) ? 1.0 / F[8] : 1.0 / std::sqrt(nrm);
  for (int a = 0; a < 9; a++) F[a] *= sc;
  return true;
}

// larger of the two squared point-to-epipolar-line distances
double residual(const double F[9], const Pt& a, const Pt& b) {
  const double l2x = F[0] * a.x + F[1] * a.y + F[2], l2y = F[3] * a.x + F[4] * a.y + F[5], l2c = F[6] * a.x + F[7] * a.y + F[8];
  const double e2 = b.x * l2x + b.y * l2y + l2c;
  const double d2 = e2 * e2 / (l2x * l2x + l2y * l2y);
  const double l1x = F[0] * b.x + F[3] * b.y + F[6], l1y = F[1] * b.x + F[4] * b.y + F[7], l1c = F[2] * b.x + F[5] * b.y + F[8];
  const double e1 = a.x * l1x + a.y * l1y + l1c;
  const double d1 = e1 * e1 / (l1x * l1x + l1y * l1y);
  const double d = d1 > d2 ? d1 : d2;
  return std::isfinite(d) ? d : 1e300;
}

bool lmeds(const std::vector<Pt>& p1, const std::vector<Pt>& p2, uint64_t seed, double F[9], uint32_t* n_inliers) {
  const int n = (int)p1.size();
  std::vector<int> all(n);
  for (int k = 0; k < n; k++) all[k] = k;
  std::vector<double> err(n);
  Rng rng{seed};
  double best_med = 1e300, bestF[9];
  bool have = false;
  for (int it = 0; it < kIterations; it++) {
    int idx[kSample];
    for (int k = 0; k < kSample;) {  // distinct indices
      const int c = (int)rng.below((uint32_t)n);
      bool dup = false;
      for (int m = 0; m < k; m++) dup |= idx[m] == c;
      if (!dup) idx[k++] = c;
    }
    double Fc[9];
    if (!eight_point(p1.data(), p2.data(), idx, kSample, Fc)) continue;
    for (int k = 0; k < n; k++) err[k] = residual(Fc, p1[k], p2[k]);
    std::nth_element(err.begin(), err.begin() + n / 2, err.end());
    const double med = err[n / 2];
    if (med < best_med) {
      best_med = med;
      memcpy(bestF, Fc, sizeof bestF);
      have = true;
    }
  }
  if (!have) return false;
  const double sigma = 2.5 * 1.4826 * (1.0 + 5.0 / (n - kSample + (n == kSample))) * std::sqrt(best_med);
  const double thr = std::max(sigma * sigma, 1e-12);
  std::vector<int> in;
  for (int k = 0; k < n; k++)
    if (residual(bestF, p1[k], p2[k]) <= thr) in.push_back(k);
  if (n_inliers) *n_inliers = (uint32_t)in.size();
  if ((int)in.size() >= kSample) {
    double Fr[9];
    if (eight_point(p1.data(), p2.data(), in.data(), (int)in.size(), Fr)) {
      // keep the refit only if it does not make the median worse
      for (int k = 0; k < n; k++) err[k] = residual(Fr, p1[k], p2[k]);
      std::nth_element(err.begin(), err.begin() + n / 2, err.end());
      if (err[n / 2] <= best_med) memcpy(bestF, Fr, sizeof bestF);
    }
  }
  memcpy(F, bestF, sizeof bestF);
  return true;
}

// per view: ascending ids of the points seen from it (get_point_sets_on_images)
std::vector<std::vector<uint32_t>> points_on_views(int V, uint64_t N, const uint32_t* off, const int32_t* view) {
  std::vector<std::vector<uint32_t>> pv((size_t)V);
  for (uint64_t p = 0; p < N; p++)
    for (uint32_t k = off[p]; k < off[p + 1]; k++) {
      const int32_t v = view[k];
      if (v < 0 || v >= V) continue;
      auto& l = pv[(size_t)v];
      if (l.empty() || l.back() != (uint32_t)p) l.push_back((uint32_t)p);  // a repeated view id counts once
    }
  return pv;
}

// get_2d_coordinates_of_point_on_image: the last listed observation with that view id
Pt obs_of(const uint32_t* off, const int32_t* view, const float* xy, uint32_t p, int v) {
  Pt r{0, 0};
  for (uint32_t k = off[p]; k < off[p + 1]; k++)
    if (view[k] == v) r = Pt{(double)xy[2 * k], (double)xy[2 * k + 1]};
  return r;
}

}  // namespace

extern "C" int eg3d_host_estimate_F(int n_views, uint64_t n_points, const uint32_t* trk_off, const int32_t* trk_view,
                                    const float* trk_xy, int estimate, uint64_t rng_seed, double* F, uint8_t* F_valid,
                                    uint32_t* n_common) {
  if (n_views <= 0 || !trk_off || !trk_view || !trk_xy || !F_valid || (estimate && !F)) return -1;
  const int V = n_views;
  const auto pv = points_on_views(V, n_points, trk_off, trk_view);
  std::vector<uint32_t> both;
  std::vector<Pt> p1, p2;
  int failed = 0;
#pragma omp parallel for schedule(dynamic) firstprivate(both, p1, p2) reduction(+ : failed)
  for (int i = 0; i < V; i++)
    for (int j = 0; j < V; j++) {
      const size_t ij = (size_t)i * V + j;
      F_valid[ij] = 0;
      if (n_common) n_common[ij] = 0;
      if (estimate) memset(F + ij * 9, 0, sizeof(double) * 9);
      if (i == j) continue;
      both.clear();
      std::set_intersection(pv[i].begin(), pv[i].end(), pv[j].begin(), pv[j].end(), std::back_inserter(both));
      if (n_common) n_common[ij] = (uint32_t)both.size();
      if ((int)both.size() < kMinCommon) continue;
      if (!estimate) {
        F_valid[ij] = 1;
        continue;
      }
      p1.clear();
      p2.clear();
      for (uint32_t p : both) {
        p1.push_back(obs_of(trk_off, trk_view, trk_xy, p, i));
        p2.push_back(obs_of(trk_off, trk_view, trk_xy, p, j));
      }
      if (lmeds(p1, p2, rng_seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(ij + 1)), F + ij * 9, nullptr))
        F_valid[ij] = 1;
      else
        failed++;
    }
  return failed;
}
