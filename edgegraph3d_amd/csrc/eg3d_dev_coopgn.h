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
#ifndef EG3D_GN_PACK_MAX
#define EG3D_GN_PACK_MAX 32      /* requests of up to this many rows are packed side by side, one chunk each */
#define EG3D_GN_ROW_CYCLES 1400  /* cost model of the longer ones: one row's projection + Jacobian (8 FP64 divisions) */
#define EG3D_GN_SUM_CYCLES 44    /*   and one row's share of the in-order sums of both passes */
#endif
// requests a window holds (request j on lane j < EG3D_COOP_REQ). 32 (round 4; it was 64): the request table halves, and
// with the per-group sums aliased onto the product columns CoopLds fits 8 LDS allocation units = FOUR single-wave
// workgroups per SIMD (EG3D_K3B_WAVES=4) instead of three. Measured neutral on its own (C3' K3b 53.7 vs 53.7 ms).
#ifndef EG3D_COOP_REQ
#define EG3D_COOP_REQ 32
#endif
// First iteration (0-based) of a Gauss-Newton solve that checks convergence on the residuals alone before it computes the
// Jacobian and the normal equations (gn_round's PRE_IT; 30 = never, the round 1-5 form). Measured (round 6,
// profiles/r06_experiments/gn_precheck_ab.txt): the many-views build gains (C4 step 1773 -> 1706 ms: its solves of ~75
// rows are bound by row arithmetic), the small / general builds do not (C3' 43.6-43.9 -> 43.8-44.2 ms, C2 6.67 -> 6.66:
// their rounds are bound by the dependent chain load -> row -> LDS -> ordered sum, which the shorter row does not
// shorten, and the extra code costs 7-18 spilled registers) — so it is on for SCENE 2 only (TeamWaveT::kPreIt).
#ifndef EG3D_GN_PRECHECK_IT
#define EG3D_GN_PRECHECK_IT 2
#endif
#ifndef EG3D_K3B_HOIST
#define EG3D_K3B_HOIST 1 /* k3b_expand's one-chunk rounds load a lane's observation once, before the iterations (gn_round's HOIST) instead of per iteration (0: rounds 4-5, when it measured as a loss; on the round-6 kernel: C3' 38.6 -> 38.2 ms, C2 4.23 -> 4.16, three runs each) */
#endif
#ifndef EG3D_GN_PRECHECK_ALL
#define EG3D_GN_PRECHECK_ALL 0 /* 1 = every build of the expand kernel, not only the many-views one (A/B switch) */
#endif
#define EG3D_STAGE_VTX 512
#define EG3D_STAGE_EPI 192
// ---- the 2-view DLT on a GROUP OF 8 LANES (round 6) ---------------------------------------------------------------
// The one-lane decomposition (svd4_smallest_v_mem) is a stream of ~8 000 vector instructions — ~230 per Jacobi
// rotation, ~36 rotations — that a wavefront executes for ONE useful lane (a uniform section) or for the few
// lanes of a look-ahead round: measured with the light timing build it was 19.5 % of k3b_expand's chain clocks on
// C3' (profiles/r06_experiments/sections_light_c3.txt). Here lane k of the group holds ROW k of A (the four entries
// At[0..3][k]) and, for k < 4, row k of V, in registers: a rotation's update of both columns is then ONE pass over
// the lanes (6 instructions instead of 84 with their LDS loads and stores), and only what the arithmetic contract
// orders — the sums over k of the dot product p and of the new squared norms a, b, which every sequential
// implementation adds k = 0, 1, ... — goes through LDS: the lanes store their products, every lane of the group
// reads them back and adds them in that order, starting from +0.0 as the sequential loop does. Every lane of the
// group thus holds the same p, a, b, W[] and takes the same decisions (skip test, rotation angle, convergence):
// same operations on the same operands in the same order as svd4_smallest_v => the same bits
// (tests/test_gpu_arith.py compares the two forms on the seeds' triples, degenerate pairs included).
// Eight groups per wavefront: the look-ahead rounds of chain following run their <= 8 DLTs side by side.
struct alignas(16) DltGrpLds {
  double s[8][4][8];  // [group][staging row][lane of the group]
};
// `on` = this lane's group has a DLT to do (uniform within a group of 8 lanes; the operands are too). Must be called by
// all 64 lanes of the single-wave block. Every lane of an `on` group returns the start point.
__device__ __forceinline__ void dlt2_grp8(DltGrpLds& S, bool on, const float* P1, float x1, float y1, const float* P2,
                                          float x2, float y2, double X0[3]) {
  constexpr int M = EG3D_DLT_M, R = EG3D_DLT_ROWS;
  const int lane = (int)(threadIdx.x & 63u), g = lane >> 3, l = lane & 7;
  double(*sg)[8] = S.s[g];
  double ar[4] = {0, 0, 0, 0}, vr[4] = {0, 0, 0, 0};
  if (on && l < M) {
    const bool second = l >= R;
    const int r = second ? l - R : l;
    const float* P = second ? P2 : P1;
    const double x = second ? x2 : x1, y = second ? y2 : y1;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (r == 0)
        ar[k] = x * (double)P[8 + k] - (double)P[k];
      else if (r == 1)
        ar[k] = y * (double)P[8 + k] - (double)P[4 + k];
      else
        ar[k] = x * (double)P[4 + k] - y * (double)P[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) vr[k] = (k == l) ? 1.0 : 0.0;
  const double eps = 2.2204460492503131e-16 * 10;
  double W[4];
  // W[i] = sum over k of At[i][k]^2, k ascending from +0.0
  {
    __syncthreads();
    if (l < M) {
#pragma unroll
      for (int i = 0; i < 4; i++) sg[i][l] = ar[i] * ar[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double sd = 0;
#pragma unroll
      for (int k = 0; k < M; k++) sd += sg[i][k];
      W[i] = sd;
    }
  }
  bool live = on;
  // one rotation of columns (i, j): static indices (the columns live in registers)
  auto rotate = [&](double& ai, double& aj, double& vi, double& vj, double& Wi, double& Wj, bool& changed) {
    __syncthreads();
    if (l < M) sg[0][l] = ai * aj;
    __syncthreads();
    double a = Wi, p = 0, b = Wj;
#pragma unroll
    for (int k = 0; k < M; k++) p += sg[0][k];
    bool skip;
    {
      const double ab = a * b, p2 = p * p, t = (eps * eps) * ab;
      if (t > 1e-250 && p2 > t * 1.0000001)
        skip = false;
      else if (t > 1e-250 && p2 < t * 0.9999999)
        skip = true;
      else
        skip = absd(p) <= eps * EG3D_SQRT(ab);
    }
    const bool act = live && !skip;
    if (!__any(act)) return;  // wave-uniform
    double c = 1, sn = 0;
    if (act) {
      p *= 2;
      double beta = a - b, gamma = EG3D_SQRT(p * p + beta * beta);
      if (beta < 0) {
        double delta = (gamma - beta) * 0.5;
        sn = EG3D_SQRT(delta / gamma);
        c = p / (gamma * sn * 2);
      } else {
        c = EG3D_SQRT((gamma + beta) / (gamma * 2));
        sn = p / (gamma * c * 2);
      }
    }
    const double t0 = c * ai + sn * aj;
    const double t1 = c * aj - sn * ai;
    if (l < M) {
      sg[1][l] = t0 * t0;
      sg[2][l] = t1 * t1;
    }
    __syncthreads();
    if (act) {
      a = 0;
      b = 0;
#pragma unroll
      for (int k = 0; k < M; k++) {
        a += sg[1][k];
        b += sg[2][k];
      }
      ai = t0;
      aj = t1;
      Wi = a;
      Wj = b;
      changed = true;
      const double u0 = c * vi + sn * vj;
      const double u1 = c * vj - sn * vi;
      vi = u0;
      vj = u1;
    }
  };
#pragma unroll 1
  for (int iter = 0; iter < 30; iter++) {
    if (!__any(live)) break;
    bool changed = false;
    rotate(ar[0], ar[1], vr[0], vr[1], W[0], W[1], changed);
    rotate(ar[0], ar[2], vr[0], vr[2], W[0], W[2], changed);
    rotate(ar[0], ar[3], vr[0], vr[3], W[0], W[3], changed);
    rotate(ar[1], ar[2], vr[1], vr[2], W[1], W[2], changed);
    rotate(ar[1], ar[3], vr[1], vr[3], W[1], W[3], changed);
    rotate(ar[2], ar[3], vr[2], vr[3], W[2], W[3], changed);
    if (!changed) live = false;
  }
  // singular values (descending selection sort; which column ends up last)
  double Ws[4];
  {
    __syncthreads();
    if (l < M) {
#pragma unroll
      for (int i = 0; i < 4; i++) sg[i][l] = ar[i] * ar[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double sd = 0;
#pragma unroll
      for (int k = 0; k < M; k++) sd += sg[i][k];
      Ws[i] = EG3D_SQRT(sd);
    }
  }
  int order[4] = {0, 1, 2, 3};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int j = i;
#pragma unroll
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
  // out[k] = Vt[last][k]: lane k's entry `last` of its row of V
  const double mine = last == 0 ? vr[0] : last == 1 ? vr[1] : last == 2 ? vr[2] : vr[3];
  __syncthreads();
  if (l < 4) sg[0][l] = mine;
  __syncthreads();
  const float h0 = (float)sg[0][0], h1 = (float)sg[0][1], h2 = (float)sg[0][2], h3 = (float)sg[0][3];
  X0[0] = (double)(h0 / h3);
  X0[1] = (double)(h1 / h3);
  X0[2] = (double)(h2 / h3);
  __syncthreads();
}

// 9 920 bytes: gfx950 allocates LDS in 1 280-byte units, 8 units per wave = 16 single-wave
// workgroups per CU (4 per SIMD) in 160 KiB. (Rounds 1-3: 12 784 bytes, 3 per SIMD.)
struct CoopLds {
  union {
    double prod[14][EG3D_COOP_ROWS + 1];  // the rows' products of one chunk (odd stride)
    // side-walk staging (never live at the same time as a solve): the polyline being walked and
    // the epipolar lines of the chain points ahead
    struct {
      f2 vtx[EG3D_STAGE_VTX];
      float epi[EG3D_STAGE_EPI][4];
    } walk;
    // the matrices of up to 8 concurrent 2-view DLTs (look-ahead following: one per lane; a uniform section: slot 0)
    double dlt_work[8][EG3D_DLT_WORK_DOUBLES];
    // ... or, lane-group form (dlt2_grp8 below): per group of 8 lanes four staging rows for the in-order sums
    DltGrpLds dltg;
  };
  int32_t la_m[8];               // look-ahead following: observations kept by step j
  uint32_t la_fl[8];             //   and the diagnostic flags its walks raised
  int32_t la_st[8];              //   and the starting observation that produced it (the redo of a failed step goes on from there)
  Obs tmp_a[EG3D_COOP_ROWS];     // the N-view step's candidate observations (Chain::tmp_a) when they fit
  float x0[EG3D_COOP_REQ][3];    // in: start point of request j; out: its result
  const Obs* gbase[EG3D_COOP_REQ];   // observation array of request j
  int32_t ex_view[EG3D_COOP_REQ];    // the extra (ADD) observation of request j
  float ex_x[EG3D_COOP_REQ], ex_y[EG3D_COOP_REQ];
  uint16_t n16[EG3D_COOP_REQ];   // rows of request j | has-extra << 15
  uint8_t row_req[EG3D_COOP_ROWS];   // row (short rounds) / group slot (long rounds) -> request lane
  uint8_t row_k[EG3D_COOP_ROWS];     // row -> its index in the request
  uint8_t res_ok[EG3D_COOP_REQ];
  uint8_t cams_mid_range;  // DevScene::cams_mid_range (set once per workgroup): enables the shared-reciprocal rows
  uint8_t long_refused;    // a request of more than EG3D_GN_PACK_MAX rows reached a build without the long-request path
  uint32_t t_start;        // k3b_expand: low word of the constant-rate clock when the chain started (Counters::max_chain_ticks)
  // Per-group sums (G >= 2 => <= 32 groups x 7 doubles). They live in product columns 6..13: a pass's sums are written
  // after the last chunk's products have been consumed (behind its barrier), pass 2 only writes columns 0..5, and
  // every lane has read the sums before the next pass writes products again — never live together.
  __device__ __forceinline__ double& gsum(int g, int e) { return (&prod[6][0])[g * 7 + e]; }
};
static_assert(32 * 7 <= 8 * (EG3D_COOP_ROWS + 1), "the per-group sums must fit product columns 6..13");
#if EG3D_COOP_REQ <= 32
static_assert(sizeof(CoopLds) <= 10240, "CoopLds must fit 8 LDS allocation units (4 waves per SIMD)");
#else
static_assert(sizeof(CoopLds) <= 12800, "CoopLds must fit 10 LDS allocation units (3 waves per SIMD)");
#endif


// ---------------------------------------------------------------------------------------------
// Lane-GROUP Gauss-Newton: the one solver of the expand stage. A window of up to 64 requests
// (request j held by lane j: observation array `base`, nblock observations, an optional extra one,
// start point X0) is solved in ROUNDS; in a round every member request owns a GROUP of G adjacent
// lanes and its rows (observations) are dealt to them in chunks of G: a lane computes the
// projection, residuals and Jacobian rows of ITS row of the chunk, the products go to LDS and the
// group's first lanes add them to the request's accumulators in observation order, carrying the
// accumulators from chunk to chunk — every accumulator sees exactly the additions, in exactly the
// order, of the sequential solver (eg3d_dev_tri.h gauss_newton_f64) => bit-identical results
// whatever the grouping.
//   * requests of <= 64 rows: G = the request's row count (one chunk; the rows stay in registers
//     between the two passes of an iteration), requests packed side by side in lane order until
//     the 64 lanes are full;
//   * longer requests (V = 200 scenes: points carry ~74 observations): G = the largest power of
//     two with G * #long requests <= 64, several chunks per pass, the update pass recomputes the
//     rows (measured as fast as keeping 8 doubles per observation in lane-private scratch memory,
//     which is what the retired one-lane-per-solve path did — at 1 KB of scratch per lane, the
//     main source of the expand kernel's former HBM write traffic).
// Each round iterates until all of its groups have converged.
// Value of lane `src` (wave-uniform) in every lane: v_readlane_b32 through a scalar register — no trip through the LDS
// crossbar that the general __shfl (ds_bpermute_b32) takes. All lanes must be active at the call.
__device__ __forceinline__ int lane_bcast(int v, int src) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}
__device__ __forceinline__ uint32_t lane_bcast(uint32_t v, int src) { return (uint32_t)lane_bcast((int)v, src); }
__device__ __forceinline__ float lane_bcast(float v, int src) { return __int_as_float(lane_bcast(__float_as_int(v), src)); }
__device__ __forceinline__ double lane_bcast(double v, int src) {
  return __hiloint2double(lane_bcast(__double2hiint(v), src), lane_bcast(__double2loint(v), src));
}
__device__ __forceinline__ unsigned long long lane_bcast(unsigned long long v, int src) {
  return ((unsigned long long)(uint32_t)lane_bcast((int)(v >> 32), src) << 32) | (uint32_t)lane_bcast((int)(v & 0xffffffffull), src);
}
// Inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts (6 vector instructions, no trip
// through the LDS crossbar; the __shfl_up ladder is 6 dependent ds_bpermute_b32). Integers: exact whatever the order.
// All lanes must be active at the call.
__device__ __forceinline__ int wave_incl_scan(int v) {
  // within each row of 16 lanes (lanes shifted in from outside the row read 0: old = 0, bound_ctrl off)
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  // across rows: the last lane of row 0 / 2 into rows 1 / 3, then lane 31 into rows 2 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}
struct GnRow {
  double j00, j01, j02, j10, j11, j12, r0, r1;
};
#if defined(EG3D_SECTION_TIMING) && !defined(EG3D_ONE_SECTION)
#define EG3D_GN_COUNTERS 1
#endif
#if defined(EG3D_GN_COUNTERS)
// diagnostic (timing builds): [0..31] requests by iterations run, [32..63] rounds by iterations run,
// [64] requests, [65] accepted, [66] rows of all requests, [67] row-iterations a round's lanes were held
// (64 x iterations x chunks), [68] row-iterations of live groups (rows x iterations), [69] rounds,
// [70..101] LONG requests by iterations, [102] long requests, [103] long rounds
__device__ unsigned long long g_gn_dbg[128];
#define EG3D_GN_DBG(i, v) atomicAdd(&g_gn_dbg[i], (unsigned long long)(v))
// [104..110] shader-clock ticks inside gn_round: rows (+ their product stores), barrier, pass-1 sums, gsum + inverse,
// pass-2 products, pass-2 sums, update; [111] whole rounds; [112] coop_gn_groups incl. its set-up
#define EG3D_GN_T0() unsigned long long gt_ = __builtin_readcyclecounter(), gtn_
#define EG3D_GN_T(i) (gtn_ = __builtin_readcyclecounter(), gts_[i] += gtn_ - gt_, gt_ = gtn_)
#else
#define EG3D_GN_DBG(i, v) ((void)0)
#define EG3D_GN_T0() ((void)0)
#define EG3D_GN_T(i) ((void)0)
#endif
// ---- the eight divisions of a row with TWO shared reciprocals -------------------------------------------------
// A correctly rounded FP64 division is, on gfx950, the sequence  sd = div_scale(den), sn = div_scale(num),
// r = rcp(sd), two Newton steps on r (4 fma), q = sn * r, t = fma(-sd, q, sn), q' = div_fmas(t, r, q),
// div_fixup(q', den, num)  — 11 VALU instructions of which SIX depend on the divisor only. A row divides two numbers
// by zH and six by zH^2. Whenever div_scale leaves its operands alone (no operand or quotient near the ends of the
// exponent range) and no operand is NaN / infinite / zero-divisor, the sequence reduces to rcp + 4 fma per DIVISOR and
// mul + 2 fma + div_fixup per NUMERATOR, with exactly the values the full sequence produces => the same bits. That regime is
// guaranteed by a range test per row: xH, yH, zH in [2^-100, 2^100] and — checked once per scene on the host — every
// non-zero camera entry in that range too; a non-zero numerator p_a zH - p_b xH is then a difference of two
// rounded products of magnitude >= 2^-200, hence >= 2^-253, and a zero numerator gives the signed zero the full
// sequence gives. Rows outside the regime take the plain divisions. (tests/test_gpu_arith.py compares the two on
// random and edge operands; every parity test runs through it.)
struct GnRecip {
  double d, r;
};
__device__ __forceinline__ bool gn_mid_range(double v) {  // 2^-100 <= |v| < 2^101, finite
  const uint32_t e = ((uint32_t)__double2hiint(v) >> 20) & 0x7ffu;
  return (e - 923u) <= 200u;
}
__device__ __forceinline__ GnRecip gn_recip(double den) {
  double r = __builtin_amdgcn_rcp(den);
  double e = __builtin_fma(-den, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-den, r, 1.0);
  r = __builtin_fma(r, e, r);
  GnRecip R;
  R.d = den;
  R.r = r;
  return R;
}
__device__ __forceinline__ double gn_div(double num, const GnRecip& R) {
  const double q = num * R.r;
  const double t = __builtin_fma(-R.d, q, num);
  // the full sequence's last step: leaves a finite quotient alone and gives a zero numerator its sign (-0 / d)
  return __builtin_amdgcn_div_fixup(__builtin_fma(t, R.r, q), R.d, num);
}
__device__ __forceinline__ void gn_row(const float* __restrict__ P, float ox, float oy, const double X[3], GnRow& w,
                                       bool cams_mid_range = false) {
  const double p00 = P[0], p01 = P[1], p02 = P[2], p03 = P[3];
  const double p10 = P[4], p11 = P[5], p12 = P[6], p13 = P[7];
  const double p20 = P[8], p21 = P[9], p22 = P[10], p23 = P[11];
  const double xH = ((p00 * X[0] + p01 * X[1]) + p02 * X[2]) + p03 * 1.0;
  const double yH = ((p10 * X[0] + p11 * X[1]) + p12 * X[2]) + p13 * 1.0;
  const double zH = ((p20 * X[0] + p21 * X[1]) + p22 * X[2]) + p23 * 1.0;
  const double zz = zH * zH;
  if (cams_mid_range && gn_mid_range(xH) && gn_mid_range(yH) && gn_mid_range(zH)) {
    const GnRecip rz = gn_recip(zH), rzz = gn_recip(zz);
    w.r0 = (double)ox - gn_div(xH, rz);
    w.r1 = (double)oy - gn_div(yH, rz);
    w.j00 = gn_div(p00 * zH - p20 * xH, rzz);
    w.j10 = gn_div(p10 * zH - p20 * yH, rzz);
    w.j01 = gn_div(p01 * zH - p21 * xH, rzz);
    w.j11 = gn_div(p11 * zH - p21 * yH, rzz);
    w.j02 = gn_div(p02 * zH - p22 * xH, rzz);
    w.j12 = gn_div(p12 * zH - p22 * yH, rzz);
    return;
  }
  w.r0 = (double)ox - xH / zH;
  w.r1 = (double)oy - yH / zH;
  w.j00 = (p00 * zH - p20 * xH) / zz;
  w.j10 = (p10 * zH - p20 * yH) / zz;
  w.j01 = (p01 * zH - p21 * xH) / zz;
  w.j11 = (p11 * zH - p21 * yH) / zz;
  w.j02 = (p02 * zH - p22 * xH) / zz;
  w.j12 = (p12 * zH - p22 * yH) / zz;
}

// The residuals of a row alone — exactly the r0 / r1 gn_row computes (same operations on the same operands), without the
// six Jacobian entries (one shared reciprocal and 6 x (2 products, a difference, a division) less: ~45 % of a full row).
// For the convergence pre-check of gn_round.
__device__ __forceinline__ void gn_row_res(const float* __restrict__ P, float ox, float oy, const double X[3], double& r0,
                                           double& r1, bool cams_mid_range) {
  const double p00 = P[0], p01 = P[1], p02 = P[2], p03 = P[3];
  const double p10 = P[4], p11 = P[5], p12 = P[6], p13 = P[7];
  const double p20 = P[8], p21 = P[9], p22 = P[10], p23 = P[11];
  const double xH = ((p00 * X[0] + p01 * X[1]) + p02 * X[2]) + p03 * 1.0;
  const double yH = ((p10 * X[0] + p11 * X[1]) + p12 * X[2]) + p13 * 1.0;
  const double zH = ((p20 * X[0] + p21 * X[1]) + p22 * X[2]) + p23 * 1.0;
  if (cams_mid_range && gn_mid_range(xH) && gn_mid_range(yH) && gn_mid_range(zH)) {
    const GnRecip rz = gn_recip(zH);
    r0 = (double)ox - gn_div(xH, rz);
    r1 = (double)oy - gn_div(yH, rz);
    return;
  }
  r0 = (double)ox - xH / zH;
  r1 = (double)oy - yH / zH;
}

// s + A[0] + B[0] + A[1] + B[1] + ... in exactly that order (the order contract of the normal equations), with the
// LDS reads of up to 8 rows in flight before the first addition needs one: the plain loop waited out the full LDS
// latency once per row (ds_read2_b64 -> s_waitcnt lgkmcnt(0) -> two dependent adds: ~130 cycles per row, a third of
// a Gauss-Newton iteration); same additions, same order => same bits.
__device__ __forceinline__ double ordered_sum2(double s, const double* __restrict__ A, const double* __restrict__ B, int rows) {
  int m = 0;
  for (; m + 8 <= rows; m += 8) {
    double a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      a[k] = A[m + k];
      b[k] = B[m + k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      s += a[k];
      s += b[k];
    }
  }
  if (m + 4 <= rows) {
    double a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      a[k] = A[m + k];
      b[k] = B[m + k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      s += a[k];
      s += b[k];
    }
    m += 4;
  }
  if (m + 2 <= rows) {
    const double a0 = A[m], b0 = B[m], a1 = A[m + 1], b1 = B[m + 1];
    s += a0;
    s += b0;
    s += a1;
    s += b1;
    m += 2;
  }
  if (m < rows) {
    const double a0 = A[m], b0 = B[m];
    s += a0;
    s += b0;
  }
  return s;
}

// One round. Lane state: act (member of a group), l = index in its group of G >= 2 lanes starting
// at lane gb, the request's n rows (the first nb from a[], the last one the extra observation),
// start point X (identical on the lanes of a group). cmax = chunks per pass (wave-uniform maximum).
// Returns the accept flag; X holds the solution.
// KEEP = chunks of a pass whose rows stay in registers between the two passes of an iteration (16 VGPRs per chunk). 0 =
// the standard build: a one-chunk round keeps its rows, a multi-chunk round recomputes all of them in the update pass;
// 4 in the wide build for scenes with long observation lists (V >= 64), where that recomputation was a fifth of the
// kernel's time. Same values either way (a row is a function of X, which does not change between the passes).
// HOIST: a one-chunk round requests the lane's observation ONCE, before the iterations (the engine kernel: its rows come
// from the slices of many chains and miss the caches — a trip per iteration was most of a round there; k3b_expand, whose
// chain is cache-resident, measured the three extra live registers as a loss and keeps the load per iteration).
template <int KEEP, bool HOIST = false, int PRE_IT = 30>
__device__ __forceinline__ bool gn_round(const float* cam_P, CoopLds& L, bool act, int l, int G, int gb, int n, int nb,
                                         const Obs* a, int32_t xv, float xx, float xy, int cmax, double X[3]) {
  const int lane = (int)(threadIdx.x & 63u);
  int32_t hview = xv;
  float hox = xx, hoy = xy;
  if constexpr (HOIST) {
    if (cmax == 1 && act && l < nb) {
      const Obs o = a[l];
      hview = (int32_t)o.view;
      hox = o.x;
      hoy = o.y;
    }
  }
  const int gs = gb >> 1;  // slot of the group's sums (G >= 2 => distinct)
  bool done = !act, ok = false;
  double last_mse = 0;
  const double two_n = (double)(n * 2);
  int dbg_it = 0, dbg_round = 0;
#if defined(EG3D_GN_COUNTERS)
  unsigned long long gts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long gt_begin_ = __builtin_readcyclecounter();
#endif
  for (int it = 0; it < 30; it++) {
    if (!__any(!done)) break;
    dbg_round++;
    if (!done) dbg_it++;
    EG3D_GN_T0();
    // ---- convergence pre-check (round 6): from its third iteration on a solve almost always stops (C3': 92 % of the requests
    // run exactly three), and the iteration that stops needs only the mean squared residual — not the Jacobian, the twelve
    // products of the normal equations or their sums. So: residuals alone first, summed in observation order into the very
    // accumulator of pass 1 (same additions, same order => the same mse, bit for bit), the same test; a group that passes
    // it is done exactly as pass 1 would have found it, one that does not runs the full iteration below, which computes
    // the same mse again and takes the same decision (last_mse is untouched here).
    if (PRE_IT < 30 && it >= PRE_IT) {
      double macc = 0;
      const bool sums_mse = !done && l == 6 % G;
      for (int c = 0; c < cmax; c++) {
        const int r = c * G + l;
        if (!done && r < n) {
          int32_t view;
          float ox, oy;
          if (HOIST && cmax == 1) {
            view = hview;
            ox = hox;
            oy = hoy;
          } else if (r < nb) {
            view = a[r].view;
            ox = a[r].x;
            oy = a[r].y;
          } else {
            view = xv;
            ox = xx;
            oy = xy;
          }
          double r0, r1;
          gn_row_res(cam_P + (size_t)view * 16, ox, oy, X, r0, r1, L.cams_mid_range != 0);
          L.prod[12][lane] = r0 * r0;
          L.prod[13][lane] = r1 * r1;
        }
        __syncthreads();
        if (sums_mse) {
          int rows = n - c * G;
          rows = rows > G ? G : rows;
          if (rows > 0) macc = ordered_sum2(macc, &L.prod[12][gb], &L.prod[13][gb], rows);
        }
        __syncthreads();
      }
      if (sums_mse) L.gsum(gs, 6) = macc;
      __syncthreads();
      if (!done) {
        const double mse = L.gsum(gs, 6);
        if (absd(mse / two_n - last_mse) < 0.0000005) {
          done = true;
          ok = last_mse < 9;
        }
      }
      __syncthreads();
      EG3D_GN_T(0);
      if (!__any(!done)) break;
    }
    // ---- pass 1: H (6) and mse; accumulator e lives in group lane e % G, slot e / G
    double acc[4] = {0, 0, 0, 0};
    constexpr int NK = KEEP > 0 ? KEEP : 1;
    GnRow wk[NK];
#pragma unroll
    for (int q = 0; q < NK; q++) wk[q].j00 = wk[q].j01 = wk[q].j02 = wk[q].j10 = wk[q].j11 = wk[q].j12 = wk[q].r0 = wk[q].r1 = 0;
    // one chunk of pass 1: the lane's row of the chunk (kept in w), its 14 products, the in-order sums
    auto pass1_chunk = [&](int c, GnRow& w) {
      const int r = c * G + l;
      const bool rowact = !done && r < n;
      if (rowact) {
        int32_t view;
        float ox, oy;
        if (HOIST && cmax == 1) {
          view = hview;
          ox = hox;
          oy = hoy;
        } else if (r < nb) {
          view = a[r].view;
          ox = a[r].x;
          oy = a[r].y;
        } else {
          view = xv;
          ox = xx;
          oy = xy;
        }
        gn_row(cam_P + (size_t)view * 16, ox, oy, X, w, L.cams_mid_range != 0);
        L.prod[0][lane] = w.j00 * w.j00;
        L.prod[1][lane] = w.j10 * w.j10;
        L.prod[2][lane] = w.j00 * w.j01;
        L.prod[3][lane] = w.j10 * w.j11;
        L.prod[4][lane] = w.j00 * w.j02;
        L.prod[5][lane] = w.j10 * w.j12;
        L.prod[6][lane] = w.j01 * w.j01;
        L.prod[7][lane] = w.j11 * w.j11;
        L.prod[8][lane] = w.j01 * w.j02;
        L.prod[9][lane] = w.j11 * w.j12;
        L.prod[10][lane] = w.j02 * w.j02;
        L.prod[11][lane] = w.j12 * w.j12;
        L.prod[12][lane] = w.r0 * w.r0;
        L.prod[13][lane] = w.r1 * w.r1;
      }
      EG3D_GN_T(0);
      __syncthreads();
      EG3D_GN_T(1);
      if (!done) {
        int rows = n - c * G;
        rows = rows > G ? G : rows;
#pragma unroll
        for (int t = 0; t < 4; t++) {  // G >= 2 => at most 4 of the 7 accumulators per lane
          const int e = l + t * G;
          if (e < 7 && rows > 0) {
            acc[t] = ordered_sum2(acc[t], &L.prod[2 * e][gb], &L.prod[2 * e + 1][gb], rows);
          }
        }
      }
      __syncthreads();
      EG3D_GN_T(2);
    };
    if constexpr (KEEP == 0) {
      for (int c = 0; c < cmax; c++) pass1_chunk(c, wk[0]);
    } else {
#pragma unroll
      for (int q = 0; q < KEEP; q++)
        if (q < cmax) pass1_chunk(q, wk[q]);
      for (int c = KEEP; c < cmax; c++) {
        GnRow wt;
        pass1_chunk(c, wt);
      }
    }
    if (!done) {
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int e = l + t * G;
        if (e < 7) L.gsum(gs, e) = acc[t];
      }
    }
    __syncthreads();
    double I00 = 0, I01 = 0, I02 = 0, I10 = 0, I11 = 0, I12 = 0, I20 = 0, I21 = 0, I22 = 0;
    if (!done) {
      const double H00 = L.gsum(gs, 0), H01 = L.gsum(gs, 1), H02 = L.gsum(gs, 2);
      const double H11 = L.gsum(gs, 3), H12 = L.gsum(gs, 4), H22 = L.gsum(gs, 5);
      const double mse = L.gsum(gs, 6);
      if (absd(mse / two_n - last_mse) < 0.0000005) {
        done = true;
        ok = last_mse < 9;
      } else {
        last_mse = mse / two_n;
        const double H10 = H01, H20 = H02, H21 = H12;
        const double d = H00 * (H11 * H22 - H12 * H21) - H01 * (H10 * H22 - H12 * H20) + H02 * (H10 * H21 - H11 * H20);
        if (d < 0.00001) {
          done = true;
          ok = false;
        } else {
          const double id = 1. / d;
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
    __syncthreads();  // the sums are rewritten by pass 2
    EG3D_GN_T(3);
    if (!__any(!done)) break;
    // ---- pass 2: the update (H^-1 J^T) r, 3 accumulators; rows recomputed unless there is one chunk
    double dac[2] = {0, 0};
    auto pass2_chunk = [&](int c, GnRow& w, bool recompute) {
      const int r = c * G + l;
      const bool rowact = !done && r < n;
      if (rowact) {
        if (recompute) {
          int32_t view;
          float ox, oy;
          if (r < nb) {
            view = a[r].view;
            ox = a[r].x;
            oy = a[r].y;
          } else {
            view = xv;
            ox = xx;
            oy = xy;
          }
          gn_row(cam_P + (size_t)view * 16, ox, oy, X, w, L.cams_mid_range != 0);
        }
        L.prod[0][lane] = ((I00 * w.j00 + I01 * w.j01) + I02 * w.j02) * w.r0;
        L.prod[1][lane] = ((I00 * w.j10 + I01 * w.j11) + I02 * w.j12) * w.r1;
        L.prod[2][lane] = ((I10 * w.j00 + I11 * w.j01) + I12 * w.j02) * w.r0;
        L.prod[3][lane] = ((I10 * w.j10 + I11 * w.j11) + I12 * w.j12) * w.r1;
        L.prod[4][lane] = ((I20 * w.j00 + I21 * w.j01) + I22 * w.j02) * w.r0;
        L.prod[5][lane] = ((I20 * w.j10 + I21 * w.j11) + I22 * w.j12) * w.r1;
      }
      __syncthreads();
      EG3D_GN_T(4);
      if (!done) {
        int rows = n - c * G;
        rows = rows > G ? G : rows;
#pragma unroll
        for (int t = 0; t < 2; t++) {  // G >= 2 => at most 2 of the 3 accumulators per lane
          const int e = l + t * G;
          if (e < 3 && rows > 0) {
            dac[t] = ordered_sum2(dac[t], &L.prod[2 * e][gb], &L.prod[2 * e + 1][gb], rows);
          }
        }
      }
      __syncthreads();
      EG3D_GN_T(5);
    };
    if constexpr (KEEP == 0) {
      for (int c = 0; c < cmax; c++) pass2_chunk(c, wk[0], cmax > 1);
    } else {
#pragma unroll
      for (int q = 0; q < KEEP; q++)
        if (q < cmax) pass2_chunk(q, wk[q], false);
      for (int c = KEEP; c < cmax; c++) {
        GnRow wt;
        pass2_chunk(c, wt, true);
      }
    }
    if (!done) {
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int e = l + t * G;
        if (e < 3) L.gsum(gs, e) = dac[t];
      }
    }
    __syncthreads();
    if (!done) {
      X[0] += L.gsum(gs, 0);
      X[1] += L.gsum(gs, 1);
      X[2] += L.gsum(gs, 2);
    }
    __syncthreads();
    EG3D_GN_T(6);
  }
  if (act && !done) ok = last_mse < 9;
#if defined(__HIP_DEVICE_COMPILE__) && defined(EG3D_GN_COUNTERS)
  if (act && l == 0) {
    EG3D_GN_DBG(dbg_it < 31 ? dbg_it : 31, 1);
    EG3D_GN_DBG(64, 1);
    EG3D_GN_DBG(65, ok ? 1 : 0);
    EG3D_GN_DBG(66, n);
    EG3D_GN_DBG(68, n * dbg_it);
    if (cmax > 1) {
      EG3D_GN_DBG(70 + (dbg_it < 31 ? dbg_it : 31), 1);
      EG3D_GN_DBG(102, 1);
    }
  }
  if (lane == 0) {
    for (int q = 0; q < 7; q++) EG3D_GN_DBG(104 + q, gts_[q]);
    EG3D_GN_DBG(111, __builtin_readcyclecounter() - gt_begin_);
    EG3D_GN_DBG(32 + (dbg_round < 31 ? dbg_round : 31), 1);
    EG3D_GN_DBG(67, 64 * dbg_round * cmax);
    EG3D_GN_DBG(69, 1);
    if (cmax > 1) EG3D_GN_DBG(103, 1);
  }
#endif
  return ok;
}

// The solver proper: the request table (gbase, n16, ex_*, x0) of this window is ALREADY in LDS — written by
// coop_gn_groups below (request j by lane j) or by the engine kernel (eg3d_k3c_engine.h: entries written by whichever
// lanes own the requests) and made visible by a barrier. Lane j < EG3D_COOP_REQ passes want / n_req of entry j. Must be
// called by all 64 lanes; every request must have >= 2 rows. On return the table holds verdict and solution of every
// entry (L.res_ok[j], L.x0[j]) until the next call; lane j also gets its own as the return value / Xout.
template <int KEEP = 0, bool LONG_GN = true, bool HOIST = false, int PRE_IT = 30>
__device__ __forceinline__ bool coop_gn_run(const float* cam_P, CoopLds& L, bool want, int n_req, float Xout[3]) {
  const int lane = (int)(threadIdx.x & 63u);
  const bool is_short = want && n_req <= EG3D_GN_PACK_MAX;
  const bool is_long = want && n_req > EG3D_GN_PACK_MAX;
  const unsigned long long m_short = __ballot(is_short), m_long = __ballot(is_long);
  // ---- short requests: rounds of whole requests packed into <= 64 rows, in lane order
  if (m_short) {
    const int ns = is_short ? n_req : 0;
    const int pre = wave_incl_scan(ns);
    const int excl = pre - ns;
    unsigned long long todo = m_short;
    while (todo) {
      const int q0 = __ffsll((long long)todo) - 1;
      const int rbase = lane_bcast(excl, q0);
      const unsigned long long in_round = __ballot(is_short && lane >= q0 && (excl + ns - rbase) <= EG3D_COOP_ROWS);
      // contiguity: stop at the first short request >= q0 that does not fit
      const unsigned long long nofit = todo & ~in_round & ~((1ull << q0) - 1ull);
      const unsigned long long upto = nofit ? ((1ull << (__ffsll((long long)nofit) - 1)) - 1ull) : ~0ull;
      const unsigned long long members = in_round & upto & todo;
      const bool mine = (members >> lane) & 1ull;
      if (mine)
        for (int k = 0; k < ns; k++) {
          L.row_req[excl - rbase + k] = (uint8_t)lane;
          L.row_k[excl - rbase + k] = (uint8_t)k;
        }
      const int last = 63 - __builtin_clzll(members);
      const int rows = lane_bcast(excl + ns, last) - rbase;
      __syncthreads();
      const bool act = lane < rows;
      int rq = 0, l = 0, n = 2, nb = 0;
      const Obs* a = nullptr;
      int32_t xv = 0;
      float xx = 0.f, xy = 0.f;
      double X[3] = {0, 0, 0};
      if (act) {
        rq = L.row_req[lane];
        l = L.row_k[lane];
        n = L.n16[rq] & 0x7fff;
        nb = n - (L.n16[rq] >> 15);
        a = L.gbase[rq];
        xv = L.ex_view[rq];
        xx = L.ex_x[rq];
        xy = L.ex_y[rq];
        X[0] = (double)L.x0[rq][0];
        X[1] = (double)L.x0[rq][1];
        X[2] = (double)L.x0[rq][2];
      }
      const bool ok = gn_round<0, HOIST, PRE_IT>(cam_P, L, act, l, n, lane - l, n, nb, a, xv, xx, xy, 1, X);
      if (act && l == 0) {
        L.res_ok[rq] = ok ? 1 : 0;
        L.x0[rq][0] = (float)X[0];
        L.x0[rq][1] = (float)X[1];
        L.x0[rq][2] = (float)X[2];
      }
      __syncthreads();
      todo &= ~members;
    }
  }
  // ---- long requests: power-of-two groups, several chunks per pass. LONG_GN = false compiles the whole path out: the
  // kernel instantiated for scenes of at most EG3D_GN_PACK_MAX views, where no point can hold more rows than a packed
  // round takes (a point has at most one observation per view). A long request that reaches such a build anyway is
  // refused, loudly (long_refused -> CTR_LONG_REFUSED -> the call fails), never mis-solved.
  if constexpr (!LONG_GN) {
    if (m_long) L.long_refused = 1;
  }
  if (LONG_GN && m_long) {
    unsigned long long todo = m_long;
    while (todo) {
      const int Bl = __popcll(todo);
      int mxl = ((todo >> lane) & 1ull) ? n_req : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const int t = __shfl_xor(mxl, d);
        mxl = t > mxl ? t : mxl;
      }
      // group size G = 1 << lg (2..64) by a cost model in cycles: rounds x chunks x (row work, twice
      // when the update pass must recompute, + the in-order sums of a chunk's G rows in both passes)
      int lg = 6;
      {
        unsigned best = 0xffffffffu;
        for (int cand = 1; cand <= 6; cand++) {
          const int Gc = 1 << cand;
          const int k = (mxl + Gc - 1) >> cand;
          const int rounds = (Bl + (64 >> cand) - 1) / (64 >> cand);
          // rows of the chunks beyond the KEEP kept ones are computed twice per iteration (update pass)
          const int krow = KEEP == 0 ? (k > 1 ? 2 * k : k) : k + (k > KEEP ? k - KEEP : 0);
          const unsigned cost = (unsigned)rounds * ((unsigned)krow * EG3D_GN_ROW_CYCLES + (unsigned)k * EG3D_GN_SUM_CYCLES * Gc);
          if (cost < best) {
            best = cost;
            lg = cand;
          }
        }
      }
      const int per_round = 64 >> lg;
      // the first per_round requests of todo
      const int rank = __popcll(todo & ((1ull << lane) - 1ull));
      const bool mine = ((todo >> lane) & 1ull) && rank < per_round;
      const unsigned long long members = __ballot(mine);
      if (mine) L.row_req[rank] = (uint8_t)lane;
      int mxn = mine ? n_req : 0;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const int t = __shfl_xor(mxn, d);
        mxn = t > mxn ? t : mxn;
      }
      __syncthreads();
      const int G = 1 << lg, g = lane >> lg, l = lane & (G - 1);
      const bool act = g < __popcll(members);
      int rq = 0, n = 2, nb = 0;
      const Obs* a = nullptr;
      int32_t xv = 0;
      float xx = 0.f, xy = 0.f;
      double X[3] = {0, 0, 0};
      if (act) {
        rq = L.row_req[g];
        n = L.n16[rq] & 0x7fff;
        nb = n - (L.n16[rq] >> 15);
        a = L.gbase[rq];
        xv = L.ex_view[rq];
        xx = L.ex_x[rq];
        xy = L.ex_y[rq];
        X[0] = (double)L.x0[rq][0];
        X[1] = (double)L.x0[rq][1];
        X[2] = (double)L.x0[rq][2];
      }
      const bool ok = gn_round<KEEP, false, PRE_IT>(cam_P, L, act, l, G, lane - l, n, nb, a, xv, xx, xy, (mxn + G - 1) >> lg, X);
      if (act && l == 0) {
        L.res_ok[rq] = ok ? 1 : 0;
        L.x0[rq][0] = (float)X[0];
        L.x0[rq][1] = (float)X[1];
        L.x0[rq][2] = (float)X[2];
      }
      __syncthreads();
      todo &= ~members;
    }
  }
  const int tl = lane < EG3D_COOP_REQ ? lane : 0;
  Xout[0] = L.x0[tl][0];
  Xout[1] = L.x0[tl][1];
  Xout[2] = L.x0[tl][2];
  const bool res = want && L.res_ok[tl] != 0;
  __syncthreads();  // the table may be rewritten by the next window
  return res;
}

// Must be called by all 64 lanes of the (single-wave) block; every request must have >= 2 rows. On
// return lane j holds the verdict and solution of ITS request (false when !want); the request
// table keeps them too (L.res_ok[j], L.x0[j]) until the next call.
template <int KEEP = 0, bool LONG_GN = true, int PRE_IT = 30>
__device__ __forceinline__ bool coop_gn_groups(const float* cam_P, CoopLds& L, bool want_in, const Obs* base, int nblock,
                                               bool has_extra, int32_t ex_view, float ex_x, float ex_y,
                                               const float X0[3], float Xout[3]) {
  const int lane = (int)(threadIdx.x & 63u);
#if defined(EG3D_GN_COUNTERS)
  const unsigned long long gg_begin_ = __builtin_readcyclecounter();
#endif
  const bool want = want_in && lane < EG3D_COOP_REQ;  // the request table has EG3D_COOP_REQ entries (callers keep to it)
  const int n_req = want ? nblock + (has_extra ? 1 : 0) : 0;
  if (__ballot(want) == 0ull) return false;
  if (lane < EG3D_COOP_REQ) {
    L.gbase[lane] = base;
    L.n16[lane] = (uint16_t)(n_req | (has_extra ? 0x8000 : 0));
    L.ex_view[lane] = ex_view;
    L.ex_x[lane] = ex_x;
    L.ex_y[lane] = ex_y;
    L.x0[lane][0] = X0[0];
    L.x0[lane][1] = X0[1];
    L.x0[lane][2] = X0[2];
    L.res_ok[lane] = 0;
  }
  __syncthreads();
  const bool res = coop_gn_run<KEEP, LONG_GN, EG3D_K3B_HOIST != 0, PRE_IT>(cam_P, L, want, n_req, Xout);
#if defined(EG3D_GN_COUNTERS)
  if (lane == 0) EG3D_GN_DBG(112, __builtin_readcyclecounter() - gg_begin_);
#endif
  return res;
}

}  // namespace eg3d
