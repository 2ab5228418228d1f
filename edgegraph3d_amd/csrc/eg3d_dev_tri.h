// eg3d_dev_tri.h — epipolar lines, projection, 2-view DLT initialisation and FP64
// Gauss-Newton triangulation for the MI355X path (per-lane device functions).
//
// Reproduces the behaviour of the reference's em_estimate3Dpositions / em_GaussNewton /
// em_add_new_observation_to_3Dpositions (src/edgegraph3d/utils/geometry/triangulation.cpp:
// 105-323, 347-466) and of the OpenCV routines they call, under the evaluation order fixed
// in DESIGN.md "Arithmetic contract". Design differences from the reference: no matrices are
// materialised — the normal equations are accumulated in registers while streaming the
// observations twice per iteration (residual+Jacobian pass, update pass), so a solve over n
// views needs O(1) storage; observations come through a cursor so the same solver runs on
// register arrays (3-view hypotheses) and on the chain pools in HBM (expand-all-views).
#pragma once
#include "eg3d_dev_geom.h"

namespace eg3d {

// One observation of an edge-point: 16 bytes, so that a lane fetches it with ONE 128-bit load (it was
// 20: five dword loads, and a fifth more HBM traffic wherever chains do not fit the caches — the
// expand stage of the 200-view scenes moves ~1 TB per 2048 seeds). View id and polyline id share a
// word: scenes are limited to EG3D_MAX_VIEWS views and EG3D_MAX_POLYLINES_PER_VIEW polylines per view,
// checked by eg3d_create. (Bit-fields: `o.view`, `o.pl` read and assign as before.)
#define EG3D_VIEW_BITS 13
#define EG3D_MAX_VIEWS (1 << EG3D_VIEW_BITS)
#define EG3D_MAX_POLYLINES_PER_VIEW (1 << (32 - EG3D_VIEW_BITS))
struct alignas(16) Obs {
  uint32_t view : EG3D_VIEW_BITS;
  uint32_t pl : 32 - EG3D_VIEW_BITS;
  uint32_t seg;
  float x, y;
};
static_assert(sizeof(Obs) == 16, "Obs must be one 128-bit word");

// l = F_ij * (x,y,1), normalised so a^2+b^2 = 1; double accumulate, float result
// (geometric_utilities.cpp:824-843 -> cv::computeCorrespondEpilines).
EG3D_HD bool epiline(const double* F, const uint8_t* F_valid, int n_views, int from, int to, float x, float y,
                     float& la, float& lb, float& lc) {
  size_t idx = (size_t)from * n_views + to;
  const double* f = F + idx * 9;
  // the nine entries are requested together with the validity byte (one trip to memory instead of two; the
  // matrix of an invalid pair is allocated and merely unused)
  const double f0 = f[0], f1 = f[1], f2_ = f[2], f3 = f[3], f4 = f[4], f5 = f[5], f6 = f[6], f7 = f[7], f8 = f[8];
  if (!F_valid[idx]) return false;
  double t0 = x, t1 = y;
  double a = (f0 * t0 + f1 * t1) + f2_;
  double b = (f3 * t0 + f4 * t1) + f5;
  double c = (f6 * t0 + f7 * t1) + f8;
  double nu = a * a + b * b;
  nu = (nu != 0.0) ? 1. / EG3D_SQRT(nu) : 1.;
  a *= nu;
  b *= nu;
  c *= nu;
  la = (float)a;
  lb = (float)b;
  lc = (float)c;
  return true;
}

// float projection (geometric_utilities.cpp:973-977 with glm's vec4*mat4 order)
EG3D_HD void project_f32(const float* P, float X, float Y, float Z, float& u, float& v) {
  float u0 = ((P[0] * X + P[1] * Y) + P[2] * Z) + P[3] * 1.0f;
  float u1 = ((P[4] * X + P[5] * Y) + P[6] * Z) + P[7] * 1.0f;
  float u2 = ((P[8] * X + P[9] * Y) + P[10] * Z) + P[11] * 1.0f;
  u = u0 / u2;
  v = u1 / u2;
}

EG3D_HD double absd(double v) { return v < 0.0 ? -v : v; }

// Which system cv::triangulatePoints builds depends on the OpenCV release: EG3D_DLT_ROWS = 3 is the
// cvTriangulatePoints of OpenCV 2.4-3.1 (6x4: rows x*P2-P0, y*P2-P1, x*P1-y*P0 per view), 2 the later 4x4
// rewrite. The reference names OpenCV 3.1 ("OpenCV >= 3.1, tested 3.1": README.md:23,32; the calls are
// triangulation.cpp:216,290), so the DEFAULT build is the 6x4 form; libeg3d_dlt4x4.so is the same library for
// a reference linked against a later OpenCV. Compile-time here (register budget), run-time in the oracle
// (orc_set_dlt_rows); eg3d_dlt_rows() reports it.
#ifndef EG3D_DLT_ROWS
#define EG3D_DLT_ROWS 3
#endif
#define EG3D_DLT_M (2 * EG3D_DLT_ROWS)

// Smallest right singular vector of an M x 4 double matrix (given transposed: 4 rows of length M)
// by one-sided Jacobi (the algorithm OpenCV's SVD uses inside cv::triangulatePoints); rotation
// order (i<j ascending), the 30-sweep cap and the 10*DBL_EPSILON skip test are part of the
// arithmetic contract.
EG3D_HD void svd4_smallest_v(double At[4][EG3D_DLT_M], double out[4]) {
  constexpr int M = EG3D_DLT_M;
  double Vt[4][4], W[4];
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 4; k++) Vt[i][k] = (i == k) ? 1.0 : 0.0;
  const double eps = 2.2204460492503131e-16 * 10;
  for (int i = 0; i < 4; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
    W[i] = sd;
  }
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int i = 0; i < 3; i++)
      for (int j = i + 1; j < 4; j++) {
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < M; k++) p += At[i][k] * At[j][k];
        // skip test |p| <= eps*sqrt(a*b): decided on the squares whenever that is unambiguous
        // (relative margin 1e-7 >> the few-ulp error of either form), so the FP64 sqrt is only
        // evaluated in the knife-edge band — same decisions, same bits
        {
          const double ab = a * b, p2 = p * p, t = (eps * eps) * ab;
          bool skip;
          if (t > 1e-250 && p2 > t * 1.0000001)
            skip = false;
          else if (t > 1e-250 && p2 < t * 0.9999999)
            skip = true;
          else
            skip = absd(p) <= eps * EG3D_SQRT(ab);
          if (skip) continue;
        }
        p *= 2;
        double beta = a - b, gamma = EG3D_SQRT(p * p + beta * beta);
        double c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = EG3D_SQRT(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = EG3D_SQRT((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = 0;
        b = 0;
        for (int k = 0; k < M; k++) {
          double t0 = c * At[i][k] + s * At[j][k];
          double t1 = c * At[j][k] - s * At[i][k];
          At[i][k] = t0;
          At[j][k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        for (int k = 0; k < 4; k++) {
          double t0 = c * Vt[i][k] + s * Vt[j][k];
          double t1 = c * Vt[j][k] - s * Vt[i][k];
          Vt[i][k] = t0;
          Vt[j][k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < 4; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
    W[i] = EG3D_SQRT(sd);
  }
  // descending selection sort; track which row ends up last
  int order[4] = {0, 1, 2, 3};
  for (int i = 0; i < 3; i++) {
    int j = i;
    for (int k = i + 1; k < 4; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double tw = W[i];
      W[i] = W[j];
      W[j] = tw;
      int to = order[i];
      order[i] = order[j];
      order[j] = to;
    }
  }
  const int last = order[3];
  for (int k = 0; k < 4; k++) out[k] = Vt[last][k];
}

// The same decomposition with the matrices in caller-provided MEMORY (work[0 .. 4M) = At, then Vt[16], then W[4]):
// same operations in the same order => same bits. The expand kernel passes LDS: there the 80 vector registers of the
// 6x4 form's At / Vt would sit on top of a chain's whole live state and push it out to scratch memory.
template <class DP>
EG3D_HD void svd4_smallest_v_mem(DP work, double out[4]) {
  constexpr int M = EG3D_DLT_M;
  DP At = work, Vt = work + 4 * M, W = work + 4 * M + 16;
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 4; k++) Vt[i * 4 + k] = (i == k) ? 1.0 : 0.0;
  const double eps = 2.2204460492503131e-16 * 10;
  for (int i = 0; i < 4; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[i * M + k];
    W[i] = sd;
  }
#pragma unroll 1
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
#pragma unroll 1
    for (int i = 0; i < 3; i++)
#pragma unroll 1
      for (int j = i + 1; j < 4; j++) {
        // the two rows travel to registers ONCE per rotation (one memory round trip, then register arithmetic)
        double ai[M], aj[M];
        for (int k = 0; k < M; k++) {
          ai[k] = At[i * M + k];
          aj[k] = At[j * M + k];
        }
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < M; k++) p += ai[k] * aj[k];
        {
          const double ab = a * b, p2 = p * p, t = (eps * eps) * ab;
          bool skip;
          if (t > 1e-250 && p2 > t * 1.0000001)
            skip = false;
          else if (t > 1e-250 && p2 < t * 0.9999999)
            skip = true;
          else
            skip = absd(p) <= eps * EG3D_SQRT(ab);
          if (skip) continue;
        }
        p *= 2;
        double beta = a - b, gamma = EG3D_SQRT(p * p + beta * beta);
        double c, s;
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = EG3D_SQRT(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = EG3D_SQRT((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = 0;
        b = 0;
        for (int k = 0; k < M; k++) {
          double t0 = c * ai[k] + s * aj[k];
          double t1 = c * aj[k] - s * ai[k];
          At[i * M + k] = t0;
          At[j * M + k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        for (int k = 0; k < 4; k++) {
          const double vi = Vt[i * 4 + k], vj = Vt[j * 4 + k];
          double t0 = c * vi + s * vj;
          double t1 = c * vj - s * vi;
          Vt[i * 4 + k] = t0;
          Vt[j * 4 + k] = t1;
        }
      }
    if (!changed) break;
  }
  double Ws[4];
  for (int i = 0; i < 4; i++) {
    double sd = 0;
    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[i * M + k];
    Ws[i] = EG3D_SQRT(sd);
  }
  // descending selection sort; track which row ends up last
  int order[4] = {0, 1, 2, 3};
  for (int i = 0; i < 3; i++) {
    int j = i;
    for (int k = i + 1; k < 4; k++)
      if (Ws[j] < Ws[k]) j = k;
    if (i != j) {
      double tw = Ws[i];
      Ws[i] = Ws[j];
      Ws[j] = tw;
      int to = order[i];
      order[i] = order[j];
      order[j] = to;
    }
  }
  const int last = order[3];
  for (int k = 0; k < 4; k++) out[k] = Vt[last * 4 + k];
}
#define EG3D_DLT_WORK_DOUBLES (4 * EG3D_DLT_M + 16 + 4)
template <class DP>
EG3D_HD void dlt2_mem(const float* P1, float x1, float y1, const float* P2, float x2, float y2, DP work, double X0[3]) {
  constexpr int M = EG3D_DLT_M;
  {
    double x = x1, y = y1;
    for (int k = 0; k < 4; k++) {
      work[k * M + 0] = x * (double)P1[8 + k] - (double)P1[k];
      work[k * M + 1] = y * (double)P1[8 + k] - (double)P1[4 + k];
#if EG3D_DLT_ROWS == 3
      work[k * M + 2] = x * (double)P1[4 + k] - y * (double)P1[k];
#endif
    }
  }
  {
    double x = x2, y = y2;
    for (int k = 0; k < 4; k++) {
      work[k * M + EG3D_DLT_ROWS + 0] = x * (double)P2[8 + k] - (double)P2[k];
      work[k * M + EG3D_DLT_ROWS + 1] = y * (double)P2[8 + k] - (double)P2[4 + k];
#if EG3D_DLT_ROWS == 3
      work[k * M + EG3D_DLT_ROWS + 2] = x * (double)P2[4 + k] - y * (double)P2[k];
#endif
    }
  }
  double v[4];
  svd4_smallest_v_mem(work, v);
  float h0 = (float)v[0], h1 = (float)v[1], h2 = (float)v[2], h3 = (float)v[3];
  X0[0] = (double)(h0 / h3);
  X0[1] = (double)(h1 / h3);
  X0[2] = (double)(h2 / h3);
}

// 2-view DLT: per view the rows x*P(2,:)-P(0,:), y*P(2,:)-P(1,:) [, x*P(1,:)-y*P(0,:)] in double;
// the homogeneous solution is rounded to float before the float division by w
// (triangulation.cpp:216-224).
EG3D_HD void dlt2(const float* P1, float x1, float y1, const float* P2, float x2, float y2, double X0[3]) {
  double At[4][EG3D_DLT_M];  // At[k][row] = A[row][k]
  {
    double x = x1, y = y1;
    for (int k = 0; k < 4; k++) {
      At[k][0] = x * (double)P1[8 + k] - (double)P1[k];
      At[k][1] = y * (double)P1[8 + k] - (double)P1[4 + k];
#if EG3D_DLT_ROWS == 3
      At[k][2] = x * (double)P1[4 + k] - y * (double)P1[k];
#endif
    }
  }
  {
    double x = x2, y = y2;
    for (int k = 0; k < 4; k++) {
      At[k][EG3D_DLT_ROWS + 0] = x * (double)P2[8 + k] - (double)P2[k];
      At[k][EG3D_DLT_ROWS + 1] = y * (double)P2[8 + k] - (double)P2[4 + k];
#if EG3D_DLT_ROWS == 3
      At[k][EG3D_DLT_ROWS + 2] = x * (double)P2[4 + k] - y * (double)P2[k];
#endif
    }
  }
  double v[4];
  svd4_smallest_v(At, v);
  float h0 = (float)v[0], h1 = (float)v[1], h2 = (float)v[2], h3 = (float)v[3];
  X0[0] = (double)(h0 / h3);
  X0[1] = (double)(h1 / h3);
  X0[2] = (double)(h2 / h3);
}

// --- observation cursors -------------------------------------------------------
// A cursor enumerates (view, x, y) in list order and can be rewound.
struct ArrayCursor {
  const Obs* a;
  int n;          // observations in the array
  const Obs* extra;  // optional trailing observation (ADD), may be null
  int i;
  EG3D_HD void rewind() { i = 0; }
  EG3D_HD int count() const { return n + (extra ? 1 : 0); }
  EG3D_HD bool next(int32_t& view, float& x, float& y) {
    if (i < n) {
      view = a[i].view;
      x = a[i].x;
      y = a[i].y;
      i++;
      return true;
    }
    if (extra && i == n) {
      view = extra->view;
      x = extra->x;
      y = extra->y;
      i++;
      return true;
    }
    return false;
  }
};

// Observations gathered once into per-lane arrays (private memory is lane-interleaved on
// gfx950, so these reads coalesce across the wave) — used by the one-lane-per-solve batches so the
// block of observations is fetched from HBM once, not in every pass of every iteration.
#define EG3D_LOCAL_OBS 16
struct LocalCursor {
  int32_t v[EG3D_LOCAL_OBS];
  float x[EG3D_LOCAL_OBS], y[EG3D_LOCAL_OBS];
  int n, i;
  EG3D_HD void rewind() { i = 0; }
  EG3D_HD int count() const { return n; }
  EG3D_HD bool next(int32_t& view, float& ox, float& oy) {
    if (i >= n) return false;
    view = v[i];
    ox = x[i];
    oy = y[i];
    i++;
    return true;
  }
};

// FP64 Gauss-Newton over the cursor's observations from X0 (triangulation.cpp:105-176):
// <=30 iterations; stop when |mse/(2n) - last| < 5e-7; fail when det(H) < 1e-5; accept iff
// last mse < 9. H and the update are accumulated in observation order (rows 2m, 2m+1).
#ifndef EG3D_KEEP_OBS
#define EG3D_KEEP_OBS EG3D_LOCAL_OBS
#endif
template <class Cursor, int KEEP = EG3D_KEEP_OBS>
EG3D_HD bool gauss_newton_f64(const float* cam_P, Cursor& cur, const double X0[3], float Xout[3]) {
  const int n = cur.count();
  double X[3] = {X0[0], X0[1], X0[2]};
  double last_mse = 0;
  const double two_n = (double)(n * 2);
  // Jacobian rows and residuals of the first pass are kept in lane-private memory (8 doubles per
  // observation, lane-interleaved => coalesced) so the update pass does not redo the projection
  // and its 8 FP64 divisions; same values, same order => same bits. Larger n recomputes.
  // (KEEP = 0: always recompute — the expand kernel's rare per-lane fallback, which must not add a
  // kilobyte of scratch per lane to that kernel)
  double keep[KEEP > 0 ? KEEP : 1][8];
  const bool stored = KEEP > 0 && n <= KEEP;
  for (int it = 0; it < 30; it++) {
    double mse = 0;
    double H00 = 0, H01 = 0, H02 = 0, H11 = 0, H12 = 0, H22 = 0;
    cur.rewind();
    int32_t view;
    float ox, oy;
    int oi = 0;
    while (cur.next(view, ox, oy)) {
      const float* P = cam_P + (size_t)view * 16;
      double p00 = P[0], p01 = P[1], p02 = P[2], p03 = P[3];
      double p10 = P[4], p11 = P[5], p12 = P[6], p13 = P[7];
      double p20 = P[8], p21 = P[9], p22 = P[10], p23 = P[11];
      double xH = ((p00 * X[0] + p01 * X[1]) + p02 * X[2]) + p03 * 1.0;
      double yH = ((p10 * X[0] + p11 * X[1]) + p12 * X[2]) + p13 * 1.0;
      double zH = ((p20 * X[0] + p21 * X[1]) + p22 * X[2]) + p23 * 1.0;
      double r0 = (double)ox - xH / zH;
      mse += r0 * r0;
      double r1 = (double)oy - yH / zH;
      mse += r1 * r1;
      double zz = zH * zH;
      double j00 = (p00 * zH - p20 * xH) / zz;
      double j10 = (p10 * zH - p20 * yH) / zz;
      double j01 = (p01 * zH - p21 * xH) / zz;
      double j11 = (p11 * zH - p21 * yH) / zz;
      double j02 = (p02 * zH - p22 * xH) / zz;
      double j12 = (p12 * zH - p22 * yH) / zz;
      H00 += j00 * j00;
      H00 += j10 * j10;
      H01 += j00 * j01;
      H01 += j10 * j11;
      H02 += j00 * j02;
      H02 += j10 * j12;
      H11 += j01 * j01;
      H11 += j11 * j11;
      H12 += j01 * j02;
      H12 += j11 * j12;
      H22 += j02 * j02;
      H22 += j12 * j12;
      if (stored) {
        double* kp = keep[oi];
        kp[0] = j00;
        kp[1] = j01;
        kp[2] = j02;
        kp[3] = j10;
        kp[4] = j11;
        kp[5] = j12;
        kp[6] = r0;
        kp[7] = r1;
      }
      oi++;
    }
    if (absd(mse / two_n - last_mse) < 0.0000005) break;
    last_mse = mse / two_n;
    const double H10 = H01, H20 = H02, H21 = H12;
    double d = H00 * (H11 * H22 - H12 * H21) - H01 * (H10 * H22 - H12 * H20) + H02 * (H10 * H21 - H11 * H20);
    if (d < 0.00001) return false;
    double id = 1. / d;
    double I00 = (H11 * H22 - H12 * H21) * id;
    double I01 = (H02 * H21 - H01 * H22) * id;
    double I02 = (H01 * H12 - H02 * H11) * id;
    double I10 = (H12 * H20 - H10 * H22) * id;
    double I11 = (H00 * H22 - H02 * H20) * id;
    double I12 = (H02 * H10 - H00 * H12) * id;
    double I20 = (H10 * H21 - H11 * H20) * id;
    double I21 = (H01 * H20 - H00 * H21) * id;
    double I22 = (H00 * H11 - H01 * H10) * id;
    double d0 = 0, d1 = 0, d2 = 0;
    if (stored) {
      for (int k = 0; k < n; k++) {
        const double* kp = keep[k];
        const double j00 = kp[0], j01 = kp[1], j02 = kp[2], j10 = kp[3], j11 = kp[4], j12 = kp[5];
        const double r0 = kp[6], r1 = kp[7];
        d0 += ((I00 * j00 + I01 * j01) + I02 * j02) * r0;
        d0 += ((I00 * j10 + I01 * j11) + I02 * j12) * r1;
        d1 += ((I10 * j00 + I11 * j01) + I12 * j02) * r0;
        d1 += ((I10 * j10 + I11 * j11) + I12 * j12) * r1;
        d2 += ((I20 * j00 + I21 * j01) + I22 * j02) * r0;
        d2 += ((I20 * j10 + I21 * j11) + I22 * j12) * r1;
      }
      X[0] += d0;
      X[1] += d1;
      X[2] += d2;
      continue;
    }
    cur.rewind();
    while (cur.next(view, ox, oy)) {
      const float* P = cam_P + (size_t)view * 16;
      double p00 = P[0], p01 = P[1], p02 = P[2], p03 = P[3];
      double p10 = P[4], p11 = P[5], p12 = P[6], p13 = P[7];
      double p20 = P[8], p21 = P[9], p22 = P[10], p23 = P[11];
      double xH = ((p00 * X[0] + p01 * X[1]) + p02 * X[2]) + p03 * 1.0;
      double yH = ((p10 * X[0] + p11 * X[1]) + p12 * X[2]) + p13 * 1.0;
      double zH = ((p20 * X[0] + p21 * X[1]) + p22 * X[2]) + p23 * 1.0;
      double r0 = (double)ox - xH / zH;
      double r1 = (double)oy - yH / zH;
      double zz = zH * zH;
      double j00 = (p00 * zH - p20 * xH) / zz;
      double j10 = (p10 * zH - p20 * yH) / zz;
      double j01 = (p01 * zH - p21 * xH) / zz;
      double j11 = (p11 * zH - p21 * yH) / zz;
      double j02 = (p02 * zH - p22 * xH) / zz;
      double j12 = (p12 * zH - p22 * yH) / zz;
      d0 += ((I00 * j00 + I01 * j01) + I02 * j02) * r0;
      d0 += ((I00 * j10 + I01 * j11) + I02 * j12) * r1;
      d1 += ((I10 * j00 + I11 * j01) + I12 * j02) * r0;
      d1 += ((I10 * j10 + I11 * j11) + I12 * j12) * r1;
      d2 += ((I20 * j00 + I21 * j01) + I22 * j02) * r0;
      d2 += ((I20 * j10 + I21 * j11) + I22 * j12) * r1;
    }
    X[0] += d0;
    X[1] += d1;
    X[2] += d2;
  }
  if (last_mse < 9) {
    Xout[0] = (float)X[0];
    Xout[1] = (float)X[1];
    Xout[2] = (float)X[2];
    return true;
  }
  return false;
}

// TRI on an observation array: DLT on (first minimal view id, LAST entry) — Q1/Q11 — then GN.
// flags gets EG3D_FLAG_DEGENERATE_DLT (16) when both DLT views coincide.
// `dlt` = the 2-view DLT to use (default: dlt2 with its matrices in registers; the expand kernel passes one that
// keeps them in LDS).
struct DltInRegisters {
  EG3D_HD void operator()(const float* P1, float x1, float y1, const float* P2, float x2, float y2, double X0[3]) const {
    dlt2(P1, x1, y1, P2, x2, y2, X0);
  }
};
template <int KEEP = EG3D_KEEP_OBS, class Dlt = DltInRegisters>
EG3D_HD bool triangulate_array(const float* cam_P, const Obs* a, int n, float Xout[3], uint32_t& flags,
                               const Dlt& dlt = Dlt()) {
  int mi = 0;
  int32_t mv = a[0].view;
  for (int i = 0; i < n; i++)
    if (a[i].view < mv) {
      mv = a[i].view;
      mi = i;
    }
  const int la = n - 1;
  if (a[mi].view == a[la].view) flags |= 16u;
  double X0[3];
  dlt(cam_P + (size_t)a[mi].view * 16, a[mi].x, a[mi].y, cam_P + (size_t)a[la].view * 16, a[la].x, a[la].y, X0);
  ArrayCursor c;
  c.a = a;
  c.n = n;
  c.extra = nullptr;
  c.i = 0;
  return gauss_newton_f64<ArrayCursor, KEEP>(cam_P, c, X0, Xout);
}

// The same solve for EXACTLY THREE observations (every point of the 3-view hypothesis stage): the three rows live in
// registers between the two passes of an iteration — the general solver keeps them in a lane-private array sized for
// 16 observations (1 KB of scratch memory per lane, the bulk of the hypothesis kernels' HBM traffic). Same
// operations in the same order as gauss_newton_f64 with n = 3 => same bits.
struct GnRow3 {
  double j00, j01, j02, j10, j11, j12, r0, r1;
};
EG3D_HD void gn_row3(const float* P, float ox, float oy, const double X[3], GnRow3& w) {
  double p00 = P[0], p01 = P[1], p02 = P[2], p03 = P[3];
  double p10 = P[4], p11 = P[5], p12 = P[6], p13 = P[7];
  double p20 = P[8], p21 = P[9], p22 = P[10], p23 = P[11];
  double xH = ((p00 * X[0] + p01 * X[1]) + p02 * X[2]) + p03 * 1.0;
  double yH = ((p10 * X[0] + p11 * X[1]) + p12 * X[2]) + p13 * 1.0;
  double zH = ((p20 * X[0] + p21 * X[1]) + p22 * X[2]) + p23 * 1.0;
  w.r0 = (double)ox - xH / zH;
  w.r1 = (double)oy - yH / zH;
  double zz = zH * zH;
  w.j00 = (p00 * zH - p20 * xH) / zz;
  w.j10 = (p10 * zH - p20 * yH) / zz;
  w.j01 = (p01 * zH - p21 * xH) / zz;
  w.j11 = (p11 * zH - p21 * yH) / zz;
  w.j02 = (p02 * zH - p22 * xH) / zz;
  w.j12 = (p12 * zH - p22 * yH) / zz;
}
EG3D_HD bool gauss_newton3_f64(const float* cam_P, const Obs* a, const double X0[3], float Xout[3]) {
  double X[3] = {X0[0], X0[1], X0[2]};
  double last_mse = 0;
  const double two_n = 6.0;
  const float* P0 = cam_P + (size_t)a[0].view * 16;
  const float* P1 = cam_P + (size_t)a[1].view * 16;
  const float* P2 = cam_P + (size_t)a[2].view * 16;
  const float x0 = a[0].x, y0 = a[0].y, x1 = a[1].x, y1 = a[1].y, x2 = a[2].x, y2 = a[2].y;
  for (int it = 0; it < 30; it++) {
    GnRow3 w0, w1, w2;
    gn_row3(P0, x0, y0, X, w0);
    gn_row3(P1, x1, y1, X, w1);
    gn_row3(P2, x2, y2, X, w2);
    double mse = 0;
    double H00 = 0, H01 = 0, H02 = 0, H11 = 0, H12 = 0, H22 = 0;
#define EG3D_GN3_ACC(w)   \
  mse += w.r0 * w.r0;     \
  mse += w.r1 * w.r1;     \
  H00 += w.j00 * w.j00;   \
  H00 += w.j10 * w.j10;   \
  H01 += w.j00 * w.j01;   \
  H01 += w.j10 * w.j11;   \
  H02 += w.j00 * w.j02;   \
  H02 += w.j10 * w.j12;   \
  H11 += w.j01 * w.j01;   \
  H11 += w.j11 * w.j11;   \
  H12 += w.j01 * w.j02;   \
  H12 += w.j11 * w.j12;   \
  H22 += w.j02 * w.j02;   \
  H22 += w.j12 * w.j12;
    EG3D_GN3_ACC(w0)
    EG3D_GN3_ACC(w1)
    EG3D_GN3_ACC(w2)
#undef EG3D_GN3_ACC
    if (absd(mse / two_n - last_mse) < 0.0000005) break;
    last_mse = mse / two_n;
    const double H10 = H01, H20 = H02, H21 = H12;
    double d = H00 * (H11 * H22 - H12 * H21) - H01 * (H10 * H22 - H12 * H20) + H02 * (H10 * H21 - H11 * H20);
    if (d < 0.00001) return false;
    double id = 1. / d;
    double I00 = (H11 * H22 - H12 * H21) * id;
    double I01 = (H02 * H21 - H01 * H22) * id;
    double I02 = (H01 * H12 - H02 * H11) * id;
    double I10 = (H12 * H20 - H10 * H22) * id;
    double I11 = (H00 * H22 - H02 * H20) * id;
    double I12 = (H02 * H10 - H00 * H12) * id;
    double I20 = (H10 * H21 - H11 * H20) * id;
    double I21 = (H01 * H20 - H00 * H21) * id;
    double I22 = (H00 * H11 - H01 * H10) * id;
    double d0 = 0, d1 = 0, d2 = 0;
#define EG3D_GN3_UPD(w)                                                  \
  d0 += ((I00 * w.j00 + I01 * w.j01) + I02 * w.j02) * w.r0;              \
  d0 += ((I00 * w.j10 + I01 * w.j11) + I02 * w.j12) * w.r1;              \
  d1 += ((I10 * w.j00 + I11 * w.j01) + I12 * w.j02) * w.r0;              \
  d1 += ((I10 * w.j10 + I11 * w.j11) + I12 * w.j12) * w.r1;              \
  d2 += ((I20 * w.j00 + I21 * w.j01) + I22 * w.j02) * w.r0;              \
  d2 += ((I20 * w.j10 + I21 * w.j11) + I22 * w.j12) * w.r1;
    EG3D_GN3_UPD(w0)
    EG3D_GN3_UPD(w1)
    EG3D_GN3_UPD(w2)
#undef EG3D_GN3_UPD
    X[0] += d0;
    X[1] += d1;
    X[2] += d2;
  }
  if (!(last_mse < 9)) return false;
  Xout[0] = (float)X[0];
  Xout[1] = (float)X[1];
  Xout[2] = (float)X[2];
  return true;
}
// TRI on exactly three observations (triangulate_array with n = 3).
EG3D_HD bool triangulate3(const float* cam_P, const Obs* a, float Xout[3], uint32_t& flags) {
  int mi = 0;
  int32_t mv = a[0].view;
  for (int i = 0; i < 3; i++)
    if (a[i].view < mv) {
      mv = a[i].view;
      mi = i;
    }
  if (a[mi].view == a[2].view) flags |= 16u;
  double X0[3];
  dlt2(cam_P + (size_t)a[mi].view * 16, a[mi].x, a[mi].y, cam_P + (size_t)a[2].view * 16, a[2].x, a[2].y, X0);
  return gauss_newton3_f64(cam_P, a, X0, Xout);
}

// First 3-subset (std::prev_permutation order of the selection mask) that triangulates, then
// greedy ADD of the remaining observations in list order (triangulation.cpp:1105-1158).
// `sel` receives the final mask; `tmp` must hold n observations.
template <int KEEP = EG3D_KEEP_OBS, class Dlt = DltInRegisters>
EG3D_HD bool triangulate_combinations(const float* cam_P, const Obs* a, int n, Obs* tmp, uint8_t* sel,
                                      float Xout[3], uint32_t& flags, const Dlt& dlt = Dlt()) {
  // enumerate 3-subsets i<j<k in the order prev_permutation visits {1,1,1,0,...}:
  // lexicographically descending masks == ascending (i,j,k) with k fastest
  bool valid = false;
  int bi = 0, bj = 0, bk = 0;
  for (int i = 0; i < n - 2 && !valid; i++)
    for (int j = i + 1; j < n - 1 && !valid; j++)
      for (int k = j + 1; k < n && !valid; k++) {
        tmp[0] = a[i];
        tmp[1] = a[j];
        tmp[2] = a[k];
        if (triangulate_array<KEEP>(cam_P, tmp, 3, Xout, flags, dlt)) {
          valid = true;
          bi = i;
          bj = j;
          bk = k;
        }
      }
  if (!valid) return false;
  for (int i = 0; i < n; i++) sel[i] = (i == bi || i == bj || i == bk) ? 1 : 0;
  tmp[0] = a[bi];
  tmp[1] = a[bj];
  tmp[2] = a[bk];
  int m = 3;
  for (int i = 0; i < n; i++) {
    if (!sel[i]) {
      ArrayCursor c;
      c.a = tmp;
      c.n = m;
      c.extra = &a[i];
      c.i = 0;
      double X0[3] = {(double)Xout[0], (double)Xout[1], (double)Xout[2]};
      float Xn[3];
      if (gauss_newton_f64<ArrayCursor, KEEP>(cam_P, c, X0, Xn)) {
        sel[i] = 1;
        Xout[0] = Xn[0];
        Xout[1] = Xn[1];
        Xout[2] = Xn[2];
        tmp[m++] = a[i];
      }
    }
  }
  return true;
}

// ------------------------------------------------------------ config 5 (FP32) ---
// One point of gaussNewtonFiltering (src/edgegraph3d/filtering/gauss_newton.cpp:83-134):
// FP32 state, products of the normal equations accumulated in double and rounded to float
// per element (OpenCV float GEMM), determinant/inverse of the 3x3 in double.
// Pointer types are template parameters so that the filter kernel can run it on operands staged in
// LDS (address_space(3) pointers => ds_read) as well as on HBM arrays; pstride = floats per camera
// matrix in cam_P (16 in HBM, 12 in the kernel's LDS copy: the last row is never read).
// The iteration is RESUMABLE: gauss_newton_f32_span runs passes [it0, it1) on a state (X, last_mse) and says whether the point
// is finished — the filter kernel runs the first passes of all its points, packs the ones still running into dense
// wavefronts and continues them; the arithmetic of a point does not depend on where its passes were cut.
struct GnF32State {
  float X[3];
  float last_mse;
};
enum { GN_F32_RUNNING = 0, GN_F32_DONE = 1, GN_F32_FAILED = -1 };
template <class PP, class VP, class XYP>
EG3D_HD int gauss_newton_f32_span(PP cam_P, int pstride, VP views, XYP xy, int n, GnF32State& st, bool legacy_abs, int it0,
                                  int it1, int* n_iter = nullptr /* diagnostics: residual passes run */) {
  float X[3] = {st.X[0], st.X[1], st.X[2]};
  float last_mse = st.last_mse;
  int result = GN_F32_RUNNING;
  for (int it = it0; it < it1; it++) {
    if (n_iter) *n_iter = it + 1;
    float mse = 0;
    double H00 = 0, H01 = 0, H02 = 0, H11 = 0, H12 = 0, H22 = 0;
    for (int m = 0; m < n; m++) {
      const PP P = cam_P + (size_t)views[m] * pstride;
      float xH = ((P[0] * X[0] + P[1] * X[1]) + P[2] * X[2]) + P[3] * 1.0f;
      float yH = ((P[4] * X[0] + P[5] * X[1]) + P[6] * X[2]) + P[7] * 1.0f;
      float zH = ((P[8] * X[0] + P[9] * X[1]) + P[10] * X[2]) + P[11] * 1.0f;
      float r0 = xy[2 * m] - xH / zH;
      mse += r0 * r0;
      float r1 = xy[2 * m + 1] - yH / zH;
      mse += r1 * r1;
      float zz = zH * zH;
      float j00 = (P[0] * zH - P[8] * xH) / zz;
      float j10 = (P[4] * zH - P[8] * yH) / zz;
      float j01 = (P[1] * zH - P[9] * xH) / zz;
      float j11 = (P[5] * zH - P[9] * yH) / zz;
      float j02 = (P[2] * zH - P[10] * xH) / zz;
      float j12 = (P[6] * zH - P[10] * yH) / zz;
      H00 += (double)j00 * (double)j00;
      H00 += (double)j10 * (double)j10;
      H01 += (double)j00 * (double)j01;
      H01 += (double)j10 * (double)j11;
      H02 += (double)j00 * (double)j02;
      H02 += (double)j10 * (double)j12;
      H11 += (double)j01 * (double)j01;
      H11 += (double)j11 * (double)j11;
      H12 += (double)j01 * (double)j02;
      H12 += (double)j11 * (double)j12;
      H22 += (double)j02 * (double)j02;
      H22 += (double)j12 * (double)j12;
    }
    float diff = mse / (float)(n * 2) - last_mse;
    bool conv;
    if (legacy_abs) {
      // Q9: ::abs(int) of the truncated difference is 0 exactly when -1 < diff < 1. A NaN or out-of-range
      // difference (a point without observations gives 0/0) is undefined behaviour in the reference's
      // float -> int conversion; x86 yields INT_MIN there, i.e. "not converged", which this predicate states
      // explicitly instead of relying on the conversion (the GPU's cvt returns 0 for NaN).
      conv = diff > -1.0f && diff < 1.0f;
    } else {
      conv = (double)EG3D_FABSF(diff) < 0.0000000005;
    }
    if (conv) {
      result = GN_F32_DONE;
      break;
    }
    last_mse = mse / (float)(n * 2);
    float h00 = (float)H00, h01 = (float)H01, h02 = (float)H02, h11 = (float)H11, h12 = (float)H12, h22 = (float)H22;
    float h10 = h01, h20 = h02, h21 = h12;
    double dd = h00 * ((double)h11 * h22 - (double)h12 * h21) - h01 * ((double)h10 * h22 - (double)h12 * h20) +
                h02 * ((double)h10 * h21 - (double)h11 * h20);
    float d = (float)dd;
    if ((double)d < 0.0000000001) return GN_F32_FAILED;
    float I00 = 0, I01 = 0, I02 = 0, I10 = 0, I11 = 0, I12 = 0, I20 = 0, I21 = 0, I22 = 0;
    if (dd != 0.) {
      double id = 1. / dd;
      I00 = (float)(((double)h11 * h22 - (double)h12 * h21) * id);
      I01 = (float)(((double)h02 * h21 - (double)h01 * h22) * id);
      I02 = (float)(((double)h01 * h12 - (double)h02 * h11) * id);
      I10 = (float)(((double)h12 * h20 - (double)h10 * h22) * id);
      I11 = (float)(((double)h00 * h22 - (double)h02 * h20) * id);
      I12 = (float)(((double)h02 * h10 - (double)h00 * h12) * id);
      I20 = (float)(((double)h10 * h21 - (double)h11 * h20) * id);
      I21 = (float)(((double)h01 * h20 - (double)h00 * h21) * id);
      I22 = (float)(((double)h00 * h11 - (double)h01 * h10) * id);
    }
    double d0 = 0, d1 = 0, d2 = 0;
    for (int m = 0; m < n; m++) {
      const PP P = cam_P + (size_t)views[m] * pstride;
      float xH = ((P[0] * X[0] + P[1] * X[1]) + P[2] * X[2]) + P[3] * 1.0f;
      float yH = ((P[4] * X[0] + P[5] * X[1]) + P[6] * X[2]) + P[7] * 1.0f;
      float zH = ((P[8] * X[0] + P[9] * X[1]) + P[10] * X[2]) + P[11] * 1.0f;
      float r0 = xy[2 * m] - xH / zH;
      float r1 = xy[2 * m + 1] - yH / zH;
      float zz = zH * zH;
      float j00 = (P[0] * zH - P[8] * xH) / zz;
      float j10 = (P[4] * zH - P[8] * yH) / zz;
      float j01 = (P[1] * zH - P[9] * xH) / zz;
      float j11 = (P[5] * zH - P[9] * yH) / zz;
      float j02 = (P[2] * zH - P[10] * xH) / zz;
      float j12 = (P[6] * zH - P[10] * yH) / zz;
      float m00 = (float)(((double)I00 * j00 + (double)I01 * j01) + (double)I02 * j02);
      float m01 = (float)(((double)I00 * j10 + (double)I01 * j11) + (double)I02 * j12);
      float m10 = (float)(((double)I10 * j00 + (double)I11 * j01) + (double)I12 * j02);
      float m11 = (float)(((double)I10 * j10 + (double)I11 * j11) + (double)I12 * j12);
      float m20 = (float)(((double)I20 * j00 + (double)I21 * j01) + (double)I22 * j02);
      float m21 = (float)(((double)I20 * j10 + (double)I21 * j11) + (double)I22 * j12);
      d0 += (double)m00 * (double)r0;
      d0 += (double)m01 * (double)r1;
      d1 += (double)m10 * (double)r0;
      d1 += (double)m11 * (double)r1;
      d2 += (double)m20 * (double)r0;
      d2 += (double)m21 * (double)r1;
    }
    X[0] += (float)d0;
    X[1] += (float)d1;
    X[2] += (float)d2;
  }
  st.X[0] = X[0];
  st.X[1] = X[1];
  st.X[2] = X[2];
  st.last_mse = last_mse;
  return result;
}
template <class PP, class VP, class XYP>
EG3D_HD bool gauss_newton_f32_t(PP cam_P, int pstride, VP views, XYP xy, int n, const float X0[3], float gn_max_mse,
                                bool legacy_abs, float Xout[3], int* n_iter = nullptr) {
  GnF32State st;
  st.X[0] = X0[0];
  st.X[1] = X0[1];
  st.X[2] = X0[2];
  st.last_mse = 0;
  if (gauss_newton_f32_span(cam_P, pstride, views, xy, n, st, legacy_abs, 0, 30, n_iter) == GN_F32_FAILED) return false;
  if (st.last_mse < gn_max_mse) {  // (converged or all 30 passes run: gauss_newton.cpp:130-133 accepts on the last mse either way)
    Xout[0] = st.X[0];
    Xout[1] = st.X[1];
    Xout[2] = st.X[2];
    return true;
  }
  return false;
}

EG3D_HD bool gauss_newton_f32(const float* cam_P, const int32_t* views, const float* xy, int n, const float X0[3],
                              float gn_max_mse, bool legacy_abs, float Xout[3], int* n_iter = nullptr) {
  return gauss_newton_f32_t(cam_P, 16, views, xy, n, X0, gn_max_mse, legacy_abs, Xout, n_iter);
}

}  // namespace eg3d
