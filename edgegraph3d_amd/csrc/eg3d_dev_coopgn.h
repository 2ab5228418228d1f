// eg3d_dev_coopgn.h — wave-cooperative FP64 Gauss-Newton for the expand stage (device only).
//
// The sequential solver (gauss_newton_f64, eg3d_dev_tri.h) spends ~8 FP64 divisions per
// observation per iteration in one lane. Here ONE OBSERVATION = ONE LANE ("row"): every row
// computes its projection, residuals and Jacobian rows at once; the normal equations are then
// summed by a few lanes of the group IN OBSERVATION ORDER through an LDS staging area, so that
// every accumulator sees exactly the additions, in exactly the order, of the sequential solver
// (triangulation.cpp:105-176 restated in eg3d_dev_tri.h) => bit-identical results.
// Several solves ("groups") are packed side by side into the 64 rows of a wavefront; each group
// converges on its own, the wave iterates until all are done.
//
// LDS per wave: 14 product columns x 65 doubles (odd stride: the summing lanes of a group read
// different columns at the same row => distinct banks) + per-group sums + the request table.
#pragma once
#include "eg3d_dev_expand.h"

namespace eg3d {

#define EG3D_COOP_ROWS 64
#define EG3D_STAGE_VTX 512
#define EG3D_STAGE_EPI 192
struct CoopLds {
  union {
    double prod[14][EG3D_COOP_ROWS + 1];
    // side-walk staging (never live at the same time as a solve): the polyline being walked and
    // the epipolar lines of the chain points ahead
    struct {
      f2 vtx[EG3D_STAGE_VTX];
      float epi[EG3D_STAGE_EPI][4];
    } walk;
  };
  int32_t la_m[8];               // look-ahead following: observations kept by step j
  uint32_t la_fl[8];             //   and the diagnostic flags its walks raised
  Obs tmp_a[EG3D_COOP_ROWS];     // the N-view step's candidate observations (Chain::tmp_a) when they fit
  double sums[22][8];            // a group has >= 3 rows => <= 21 groups per round
  float x0[EG3D_COOP_ROWS][3];   // in: start point of request j; out: its result
  uint32_t off[EG3D_COOP_ROWS];  // first observation of request j in the chain's pool
  int32_t n[EG3D_COOP_ROWS];     // rows of request j (block observations + the extra one), 0 = none
  int32_t start[EG3D_COOP_ROWS]; // exclusive prefix of n over the window
  int32_t ex_view[EG3D_COOP_ROWS];
  float ex_x[EG3D_COOP_ROWS], ex_y[EG3D_COOP_ROWS];
  uint8_t row_req[EG3D_COOP_ROWS], row_k[EG3D_COOP_ROWS], row_g[EG3D_COOP_ROWS];
  uint8_t res_ok[EG3D_COOP_ROWS];
};

// Iterations of all groups currently mapped onto the wave. A row lane passes act=true, its group
// ordinal g (index into sums), its row k within the group, the group's row count n and first
// lane gb, its observation and the start point X (identical on the rows of a group). Returns the
// accept flag; X holds the solution (identical on the rows of a group). Must be called by all 64
// lanes of the (single-wave) block.
__device__ __forceinline__ bool coop_gn_rows(const float* cam_P, CoopLds& L, bool act, int g, int k, int n, int gb,
                                             int32_t view, float ox, float oy, double X[3]) {
  const int lane = (int)(threadIdx.x & 63u);
  float pf[12];
  if (act) {
    const float* P = cam_P + (size_t)view * 16;
#pragma unroll
    for (int i = 0; i < 12; i++) pf[i] = P[i];
  } else {
#pragma unroll
    for (int i = 0; i < 12; i++) pf[i] = 0.f;
  }
  bool done = !act;
  bool ok = false;
  double last_mse = 0;
  const double two_n = (double)(n * 2);
  for (int it = 0; it < 30; it++) {
    if (!__any(!done)) break;
    double j00 = 0, j01 = 0, j02 = 0, j10 = 0, j11 = 0, j12 = 0, r0 = 0, r1 = 0;
    if (!done) {
      const double p00 = pf[0], p01 = pf[1], p02 = pf[2], p03 = pf[3];
      const double p10 = pf[4], p11 = pf[5], p12 = pf[6], p13 = pf[7];
      const double p20 = pf[8], p21 = pf[9], p22 = pf[10], p23 = pf[11];
      double xH = ((p00 * X[0] + p01 * X[1]) + p02 * X[2]) + p03 * 1.0;
      double yH = ((p10 * X[0] + p11 * X[1]) + p12 * X[2]) + p13 * 1.0;
      double zH = ((p20 * X[0] + p21 * X[1]) + p22 * X[2]) + p23 * 1.0;
      r0 = (double)ox - xH / zH;
      r1 = (double)oy - yH / zH;
      double zz = zH * zH;
      j00 = (p00 * zH - p20 * xH) / zz;
      j10 = (p10 * zH - p20 * yH) / zz;
      j01 = (p01 * zH - p21 * xH) / zz;
      j11 = (p11 * zH - p21 * yH) / zz;
      j02 = (p02 * zH - p22 * xH) / zz;
      j12 = (p12 * zH - p22 * yH) / zz;
      L.prod[0][lane] = j00 * j00;
      L.prod[1][lane] = j10 * j10;
      L.prod[2][lane] = j00 * j01;
      L.prod[3][lane] = j10 * j11;
      L.prod[4][lane] = j00 * j02;
      L.prod[5][lane] = j10 * j12;
      L.prod[6][lane] = j01 * j01;
      L.prod[7][lane] = j11 * j11;
      L.prod[8][lane] = j01 * j02;
      L.prod[9][lane] = j11 * j12;
      L.prod[10][lane] = j02 * j02;
      L.prod[11][lane] = j12 * j12;
      L.prod[12][lane] = r0 * r0;
      L.prod[13][lane] = r1 * r1;
    }
    __syncthreads();
    if (!done) {
      // H00 H01 H02 H11 H12 H22 mse: accumulator e adds (row 2m, row 2m+1) products, m ascending
      for (int e = k; e < 7; e += n) {
        const double* A = &L.prod[2 * e][gb];
        const double* B = &L.prod[2 * e + 1][gb];
        double acc = 0;
        for (int m = 0; m < n; m++) {
          acc += A[m];
          acc += B[m];
        }
        L.sums[g][e] = acc;
      }
    }
    __syncthreads();
    double I00 = 0, I01 = 0, I02 = 0, I10 = 0, I11 = 0, I12 = 0, I20 = 0, I21 = 0, I22 = 0;
    if (!done) {
      const double H00 = L.sums[g][0], H01 = L.sums[g][1], H02 = L.sums[g][2];
      const double H11 = L.sums[g][3], H12 = L.sums[g][4], H22 = L.sums[g][5];
      const double mse = L.sums[g][6];
      if (absd(mse / two_n - last_mse) < 0.0000005) {
        done = true;
        ok = last_mse < 9;
      } else {
        last_mse = mse / two_n;
        const double H10 = H01, H20 = H02, H21 = H12;
        double d = H00 * (H11 * H22 - H12 * H21) - H01 * (H10 * H22 - H12 * H20) + H02 * (H10 * H21 - H11 * H20);
        if (d < 0.00001) {
          done = true;
          ok = false;
        } else {
          double id = 1. / d;
          I00 = (H11 * H22 - H12 * H21) * id;
          I01 = (H02 * H21 - H01 * H22) * id;
          I02 = (H01 * H12 - H02 * H11) * id;
          I10 = (H12 * H20 - H10 * H22) * id;
          I11 = (H00 * H22 - H02 * H20) * id;
          I12 = (H02 * H10 - H00 * H12) * id;
          I20 = (H10 * H21 - H11 * H20) * id;
          I21 = (H01 * H20 - H00 * H21) * id;
          I22 = (H00 * H11 - H01 * H10) * id;
        }
      }
    }
    if (!done) {
      L.prod[0][lane] = ((I00 * j00 + I01 * j01) + I02 * j02) * r0;
      L.prod[1][lane] = ((I00 * j10 + I01 * j11) + I02 * j12) * r1;
      L.prod[2][lane] = ((I10 * j00 + I11 * j01) + I12 * j02) * r0;
      L.prod[3][lane] = ((I10 * j10 + I11 * j11) + I12 * j12) * r1;
      L.prod[4][lane] = ((I20 * j00 + I21 * j01) + I22 * j02) * r0;
      L.prod[5][lane] = ((I20 * j10 + I21 * j11) + I22 * j12) * r1;
    }
    __syncthreads();
    if (!done) {
      for (int e = k; e < 3; e += n) {
        const double* A = &L.prod[2 * e][gb];
        const double* B = &L.prod[2 * e + 1][gb];
        double acc = 0;
        for (int m = 0; m < n; m++) {
          acc += A[m];
          acc += B[m];
        }
        L.sums[g][e] = acc;
      }
    }
    __syncthreads();
    if (!done) {
      X[0] += L.sums[g][0];
      X[1] += L.sums[g][1];
      X[2] += L.sums[g][2];
    }
    // the next iteration's first barrier separates these reads from its sums writes
  }
  if (act && !done) ok = last_mse < 9;
  return ok;
}

// One solve shared by the whole wave (a uniform section): rows = lanes < n. a[] is readable by
// every lane; the result is returned to all lanes.
__device__ __forceinline__ bool coop_gn_single(const float* cam_P, CoopLds& L, const Obs* a, int n, const double X0[3],
                                               float Xout[3]) {
  const int lane = (int)(threadIdx.x & 63u);
  const bool act = lane < n;
  int32_t view = 0;
  float ox = 0.f, oy = 0.f;
  if (act) {
    view = a[lane].view;
    ox = a[lane].x;
    oy = a[lane].y;
  }
  double X[3] = {X0[0], X0[1], X0[2]};
  bool ok = coop_gn_rows(cam_P, L, act, 0, lane, n, 0, view, ox, oy, X);
  // lane 0 is always a row
  const int oki = __shfl(ok ? 1 : 0, 0);
  float x0 = (float)X[0], x1 = (float)X[1], x2 = (float)X[2];
  Xout[0] = __shfl(x0, 0);
  Xout[1] = __shfl(x1, 0);
  Xout[2] = __shfl(x2, 0);
  return oki != 0;
}

// ONE solve with any number of observations, all 64 lanes on it: rows are processed in chunks of
// 64; the <=7 accumulating lanes carry their sums across the chunks, so the additions still happen
// in observation order. The update pass recomputes the rows (no per-row state survives a chunk).
// Arguments are wave-uniform; `a` holds n_arr observations, an optional extra one follows them.
__device__ __forceinline__ bool coop_gn_big(const float* cam_P, CoopLds& L, const Obs* a, int n_arr, bool has_extra,
                                            int32_t ex_view, float ex_x, float ex_y, const double X0[3],
                                            float Xout[3]) {
  const int lane = (int)(threadIdx.x & 63u);
  const int n = n_arr + (has_extra ? 1 : 0);
  double X[3] = {X0[0], X0[1], X0[2]};
  double last_mse = 0;
  const double two_n = (double)(n * 2);
  bool ok = false, done = false;
  for (int it = 0; it < 30 && !done; it++) {
    double acc = 0;  // lane e < 7: H00 H01 H02 H11 H12 H22 mse
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int r = c0 + lane;
      const int rows = (n - c0) < 64 ? (n - c0) : 64;
      if (r < n) {
        int32_t view;
        float ox, oy;
        if (r < n_arr) {
          view = a[r].view;
          ox = a[r].x;
          oy = a[r].y;
        } else {
          view = ex_view;
          ox = ex_x;
          oy = ex_y;
        }
        const float* P = cam_P + (size_t)view * 16;
        const double p00 = P[0], p01 = P[1], p02 = P[2], p03 = P[3];
        const double p10 = P[4], p11 = P[5], p12 = P[6], p13 = P[7];
        const double p20 = P[8], p21 = P[9], p22 = P[10], p23 = P[11];
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
        L.prod[0][lane] = j00 * j00;
        L.prod[1][lane] = j10 * j10;
        L.prod[2][lane] = j00 * j01;
        L.prod[3][lane] = j10 * j11;
        L.prod[4][lane] = j00 * j02;
        L.prod[5][lane] = j10 * j12;
        L.prod[6][lane] = j01 * j01;
        L.prod[7][lane] = j11 * j11;
        L.prod[8][lane] = j01 * j02;
        L.prod[9][lane] = j11 * j12;
        L.prod[10][lane] = j02 * j02;
        L.prod[11][lane] = j12 * j12;
        L.prod[12][lane] = r0 * r0;
        L.prod[13][lane] = r1 * r1;
      }
      __syncthreads();
      if (lane < 7) {
        const double* A = &L.prod[2 * lane][0];
        const double* B = &L.prod[2 * lane + 1][0];
        for (int m = 0; m < rows; m++) {
          acc += A[m];
          acc += B[m];
        }
      }
      __syncthreads();
    }
    if (lane < 7) L.sums[0][lane] = acc;
    __syncthreads();
    const double H00 = L.sums[0][0], H01 = L.sums[0][1], H02 = L.sums[0][2];
    const double H11 = L.sums[0][3], H12 = L.sums[0][4], H22 = L.sums[0][5];
    const double mse = L.sums[0][6];
    __syncthreads();
    if (absd(mse / two_n - last_mse) < 0.0000005) {
      done = true;
      ok = last_mse < 9;
      break;
    }
    last_mse = mse / two_n;
    const double H10 = H01, H20 = H02, H21 = H12;
    double d = H00 * (H11 * H22 - H12 * H21) - H01 * (H10 * H22 - H12 * H20) + H02 * (H10 * H21 - H11 * H20);
    if (d < 0.00001) {
      done = true;
      ok = false;
      break;
    }
    double id = 1. / d;
    const double I00 = (H11 * H22 - H12 * H21) * id;
    const double I01 = (H02 * H21 - H01 * H22) * id;
    const double I02 = (H01 * H12 - H02 * H11) * id;
    const double I10 = (H12 * H20 - H10 * H22) * id;
    const double I11 = (H00 * H22 - H02 * H20) * id;
    const double I12 = (H02 * H10 - H00 * H12) * id;
    const double I20 = (H10 * H21 - H11 * H20) * id;
    const double I21 = (H01 * H20 - H00 * H21) * id;
    const double I22 = (H00 * H11 - H01 * H10) * id;
    double dacc = 0;  // lane e < 3: d0 d1 d2
    for (int c0 = 0; c0 < n; c0 += 64) {
      const int r = c0 + lane;
      const int rows = (n - c0) < 64 ? (n - c0) : 64;
      if (r < n) {
        int32_t view;
        float ox, oy;
        if (r < n_arr) {
          view = a[r].view;
          ox = a[r].x;
          oy = a[r].y;
        } else {
          view = ex_view;
          ox = ex_x;
          oy = ex_y;
        }
        const float* P = cam_P + (size_t)view * 16;
        const double p00 = P[0], p01 = P[1], p02 = P[2], p03 = P[3];
        const double p10 = P[4], p11 = P[5], p12 = P[6], p13 = P[7];
        const double p20 = P[8], p21 = P[9], p22 = P[10], p23 = P[11];
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
        L.prod[0][lane] = ((I00 * j00 + I01 * j01) + I02 * j02) * r0;
        L.prod[1][lane] = ((I00 * j10 + I01 * j11) + I02 * j12) * r1;
        L.prod[2][lane] = ((I10 * j00 + I11 * j01) + I12 * j02) * r0;
        L.prod[3][lane] = ((I10 * j10 + I11 * j11) + I12 * j12) * r1;
        L.prod[4][lane] = ((I20 * j00 + I21 * j01) + I22 * j02) * r0;
        L.prod[5][lane] = ((I20 * j10 + I21 * j11) + I22 * j12) * r1;
      }
      __syncthreads();
      if (lane < 3) {
        const double* A = &L.prod[2 * lane][0];
        const double* B = &L.prod[2 * lane + 1][0];
        for (int m = 0; m < rows; m++) {
          dacc += A[m];
          dacc += B[m];
        }
      }
      __syncthreads();
    }
    if (lane < 3) L.sums[0][lane] = dacc;
    __syncthreads();
    X[0] += L.sums[0][0];
    X[1] += L.sums[0][1];
    X[2] += L.sums[0][2];
    __syncthreads();
  }
  if (!done) ok = last_mse < 9;
  Xout[0] = (float)X[0];
  Xout[1] = (float)X[1];
  Xout[2] = (float)X[2];
  return ok;
}

// A window of up to 64 ADD requests, request j held by lane j: (want, block offset, block size,
// extra observation, start point). Requests are packed into rounds of <= 64 rows. On return lane
// j holds the verdict and solution of its request. Every request must have 3 <= rows <= 64.
__device__ __forceinline__ bool coop_gn_window(const float* cam_P, const Obs* pool, CoopLds& L, bool want, uint32_t off,
                                               int nblock, const Obs& extra, const float X0[3], float Xout[3]) {
  const int lane = (int)(threadIdx.x & 63u);
  const int n = want ? nblock + 1 : 0;
  int pre = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(pre, d);
    if (lane >= d) pre += t;
  }
  const int excl = pre - n;
  const int total = __shfl(pre, 63);
  L.off[lane] = off;
  L.n[lane] = n;
  L.start[lane] = excl;
  L.ex_view[lane] = extra.view;
  L.ex_x[lane] = extra.x;
  L.ex_y[lane] = extra.y;
  L.x0[lane][0] = X0[0];
  L.x0[lane][1] = X0[1];
  L.x0[lane][2] = X0[2];
  L.res_ok[lane] = 0;
  __syncthreads();
  int q0 = 0;
  while (q0 < 64) {
    const int base = L.start[q0];
    if (base >= total) break;  // only empty requests remain
    // requests [q0, q1) fit into 64 rows
    const unsigned long long fit = __ballot(lane < q0 || (excl + n - base) <= EG3D_COOP_ROWS);
    const unsigned long long nofit = ~fit;
    const int q1 = nofit ? (__ffsll((long long)nofit) - 1) : 64;
    const bool mine = lane >= q0 && lane < q1 && n > 0;
    // group ordinals: rank of this request among the non-empty ones of the round
    const unsigned long long members = __ballot(mine);
    if (mine) {
      const int g = __popcll(members & ((1ull << lane) - 1ull));
      for (int k = 0; k < n; k++) {
        L.row_req[excl - base + k] = (uint8_t)lane;
        L.row_k[excl - base + k] = (uint8_t)k;
        L.row_g[excl - base + k] = (uint8_t)g;
      }
    }
    const int rows = (q1 < 64 ? L.start[q1] : total) - base;
    __syncthreads();
    const bool act = lane < rows;
    int rq = 0, k = 0, g = 0, rn = 1;
    int32_t view = 0;
    float ox = 0.f, oy = 0.f;
    double X[3] = {0, 0, 0};
    if (act) {
      rq = L.row_req[lane];
      k = L.row_k[lane];
      g = L.row_g[lane];
      rn = L.n[rq];
      if (k < rn - 1) {
        const Obs& o = pool[L.off[rq] + k];
        view = o.view;
        ox = o.x;
        oy = o.y;
      } else {
        view = L.ex_view[rq];
        ox = L.ex_x[rq];
        oy = L.ex_y[rq];
      }
      X[0] = (double)L.x0[rq][0];
      X[1] = (double)L.x0[rq][1];
      X[2] = (double)L.x0[rq][2];
    }
    const bool ok = coop_gn_rows(cam_P, L, act, g, k, rn, lane - k, view, ox, oy, X);
    if (act && k == 0) {
      L.res_ok[rq] = ok ? 1 : 0;
      L.x0[rq][0] = (float)X[0];
      L.x0[rq][1] = (float)X[1];
      L.x0[rq][2] = (float)X[2];
    }
    __syncthreads();
    q0 = q1;
  }
  __syncthreads();
  Xout[0] = L.x0[lane][0];
  Xout[1] = L.x0[lane][1];
  Xout[2] = L.x0[lane][2];
  const bool r = want && L.res_ok[lane] != 0;
  __syncthreads();  // the table may be rewritten by the next window
  return r;
}

}  // namespace eg3d
