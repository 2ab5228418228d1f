// TEST-ONLY library (tests/probe/libeg3d_probe.so, built by edgegraph3d_amd.build.build_probe): runs
// single device primitives of the product's headers (IEEE arithmetic, DLT, triangulation) on the
// GPU so that tests/test_gpu_arith.py can check the arithmetic contract bit-for-bit against x86 and
// the oracle. Nothing of this is linked into libeg3d.so.
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "eg3d_probe.h"
#include "eg3d_dev_pipeline.h"
#include "eg3d_dev_coopgn.h"

using namespace eg3d;

__global__ void k_probe_arith(uint64_t n, const double* a, const double* b, const double* c, double* od, const float* fa,
                              const float* fb, const float* fc, float* of) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  od[0 * n + i] = a[i] / b[i];
  od[1 * n + i] = EG3D_SQRT(absd(a[i]));
  double p = a[i] * b[i];
  od[2 * n + i] = p + c[i];
  od[3 * n + i] = (double)(float)a[i];
  od[4 * n + i] = 1. / EG3D_SQRT(absd(b[i]));
  of[0 * n + i] = fa[i] / fb[i];
  of[1 * n + i] = EG3D_SQRTF(EG3D_FABSF(fa[i]));
  float q = fa[i] * fb[i];
  of[2 * n + i] = q + fc[i];
  of[3 * n + i] = dist2(fa[i], fb[i], fc[i], fa[i]);
  of[4 * n + i] = __builtin_sqrtf(EG3D_FABSF(fa[i]));
}

__global__ void k_probe_tri(const float* cam_P, uint64_t n, int k, const int32_t* views, const float* xy, float* X,
                            uint8_t* valid, double* dlt) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Obs a[16];
  for (int j = 0; j < k && j < 16; j++) {
    a[j].view = views[i * k + j];
    a[j].pl = 0;
    a[j].seg = 0;
    a[j].x = xy[2 * (i * k + j)];
    a[j].y = xy[2 * (i * k + j) + 1];
  }
  uint32_t flags = 0;
  float Xo[3] = {0, 0, 0};
  bool ok = triangulate_array(cam_P, a, k, Xo, flags);
  valid[i] = ok;
  X[3 * i] = Xo[0];
  X[3 * i + 1] = Xo[1];
  X[3 * i + 2] = Xo[2];
  int mi = 0;
  for (int j = 0; j < k; j++)
    if (a[j].view < a[mi].view) mi = j;
  double X0[3];
  dlt2(cam_P + (size_t)a[mi].view * 16, a[mi].x, a[mi].y, cam_P + (size_t)a[k - 1].view * 16, a[k - 1].x, a[k - 1].y, X0);
  dlt[3 * i] = X0[0];
  dlt[3 * i + 1] = X0[1];
  dlt[3 * i + 2] = X0[2];
}

// the 2-view DLT on groups of 8 lanes (dlt2_grp8): item i on group (i % 8) of block (i / 8); same operand choice as
// k_probe_tri (minimum view id, last observation)
__global__ void __launch_bounds__(64) k_probe_dlt_grp(const float* cam_P, uint64_t n, int k, const int32_t* views, const float* xy,
                                                      double* dlt) {
  __shared__ DltGrpLds S;
  const int lane = (int)threadIdx.x, g = lane >> 3;
  const uint64_t i = (uint64_t)blockIdx.x * 8 + (uint64_t)g;
  const bool on = i < n;
  const float* P1 = cam_P;
  const float* P2 = cam_P;
  float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
  if (on) {
    int mi = 0;
    for (int j = 0; j < k; j++)
      if (views[i * k + j] < views[i * k + mi]) mi = j;
    P1 = cam_P + (size_t)views[i * k + mi] * 16;
    x1 = xy[2 * (i * k + mi)];
    y1 = xy[2 * (i * k + mi) + 1];
    P2 = cam_P + (size_t)views[i * k + k - 1] * 16;
    x2 = xy[2 * (i * k + k - 1)];
    y2 = xy[2 * (i * k + k - 1) + 1];
  }
  double X0[3] = {0, 0, 0};
  dlt2_grp8(S, on, P1, x1, y1, P2, x2, y2, X0);
  if (on && (lane & 7) == 0) {
    dlt[3 * i] = X0[0];
    dlt[3 * i + 1] = X0[1];
    dlt[3 * i + 2] = X0[2];
  }
}

#define PT(expr)                         \
  do {                                   \
    if ((expr) != hipSuccess) return -2; \
  } while (0)

extern "C" int eg3d_probe_dlt_rows(void) { return EG3D_DLT_ROWS; }

extern "C" int eg3d_probe_arith(uint64_t n, const double* a, const double* b, const double* c, double* od,
                                const float* fa, const float* fb, const float* fc, float* of) {
  double *da, *db, *dc, *dod;
  float *dfa, *dfb, *dfc, *dof;
  PT(hipMalloc(&da, n * 8));
  PT(hipMalloc(&db, n * 8));
  PT(hipMalloc(&dc, n * 8));
  PT(hipMalloc(&dod, n * 8 * 5));
  PT(hipMalloc(&dfa, n * 4));
  PT(hipMalloc(&dfb, n * 4));
  PT(hipMalloc(&dfc, n * 4));
  PT(hipMalloc(&dof, n * 4 * 5));
  PT(hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
  PT(hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice));
  PT(hipMemcpy(dc, c, n * 8, hipMemcpyHostToDevice));
  PT(hipMemcpy(dfa, fa, n * 4, hipMemcpyHostToDevice));
  PT(hipMemcpy(dfb, fb, n * 4, hipMemcpyHostToDevice));
  PT(hipMemcpy(dfc, fc, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_arith, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, da, db, dc, dod, dfa, dfb, dfc,
                     dof);
  PT(hipDeviceSynchronize());
  PT(hipMemcpy(od, dod, n * 8 * 5, hipMemcpyDeviceToHost));
  PT(hipMemcpy(of, dof, n * 4 * 5, hipMemcpyDeviceToHost));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dc);
  (void)hipFree(dod);
  (void)hipFree(dfa);
  (void)hipFree(dfb);
  (void)hipFree(dfc);
  (void)hipFree(dof);
  return 0;
}

extern "C" int eg3d_probe_triangulate(const float* cam_P, int n_views, uint64_t n, int k, const int32_t* views,
                                      const float* xy, float* X, uint8_t* valid, double* dlt) {
  if (!cam_P || n_views < 1 || k < 2 || k > 16) return -1;
  int32_t* dv;
  float *dxy, *dX, *dP;
  PT(hipMalloc(&dP, (size_t)n_views * 64));
  PT(hipMemcpy(dP, cam_P, (size_t)n_views * 64, hipMemcpyHostToDevice));
  uint8_t* dval;
  double* ddlt;
  PT(hipMalloc(&dv, n * k * 4));
  PT(hipMalloc(&dxy, n * k * 8));
  PT(hipMalloc(&dX, n * 12));
  PT(hipMalloc(&dval, n));
  PT(hipMalloc(&ddlt, n * 24));
  PT(hipMemcpy(dv, views, n * k * 4, hipMemcpyHostToDevice));
  PT(hipMemcpy(dxy, xy, n * k * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_tri, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, dP, n, k, dv, dxy, dX, dval, ddlt);
  PT(hipDeviceSynchronize());
  PT(hipMemcpy(X, dX, n * 12, hipMemcpyDeviceToHost));
  PT(hipMemcpy(valid, dval, n, hipMemcpyDeviceToHost));
  PT(hipMemcpy(dlt, ddlt, n * 24, hipMemcpyDeviceToHost));
  (void)hipFree(dP);
  (void)hipFree(dv);
  (void)hipFree(dxy);
  (void)hipFree(dX);
  (void)hipFree(dval);
  (void)hipFree(ddlt);
  return 0;
}

extern "C" int eg3d_probe_dlt_groups(const float* cam_P, int n_views, uint64_t n, int k, const int32_t* views, const float* xy,
                                     double* dlt) {
  if (!cam_P || n_views < 1 || k < 2 || k > 16 || !n) return -1;
  int32_t* dv;
  float *dxy, *dP;
  double* ddlt;
  PT(hipMalloc(&dP, (size_t)n_views * 64));
  PT(hipMemcpy(dP, cam_P, (size_t)n_views * 64, hipMemcpyHostToDevice));
  PT(hipMalloc(&dv, n * k * 4));
  PT(hipMalloc(&dxy, n * k * 8));
  PT(hipMalloc(&ddlt, n * 24));
  PT(hipMemcpy(dv, views, n * k * 4, hipMemcpyHostToDevice));
  PT(hipMemcpy(dxy, xy, n * k * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_dlt_grp, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, 0, dP, n, k, dv, dxy, ddlt);
  PT(hipDeviceSynchronize());
  PT(hipMemcpy(dlt, ddlt, n * 24, hipMemcpyDeviceToHost));
  (void)hipFree(dP);
  (void)hipFree(dv);
  (void)hipFree(dxy);
  (void)hipFree(ddlt);
  return 0;
}

// ---- the shared-reciprocal divisions of the Gauss-Newton rows (eg3d_dev_coopgn.h) against plain divisions ----
// pairs: out[0][i] = num / den (the compiler's full sequence), out[1][i] = gn_div(num, gn_recip(den)),
// out[2][i] = 1 when den is in the range the row test admits. rows: GnRow of the plain and of the guarded fast path.
__global__ void k_probe_gn_div(uint64_t n, const double* num, const double* den, double* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = num[i] / den[i];
  const GnRecip R = gn_recip(den[i]);
  out[n + i] = gn_div(num[i], R);
  out[2 * n + i] = gn_mid_range(den[i]) ? 1.0 : 0.0;
}
__global__ void k_probe_gn_rows(uint64_t n, const float* P, const float* oxy, const double* X, double* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double Xi[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]};
  GnRow a, b;
  gn_row(P + 16 * i, oxy[2 * i], oxy[2 * i + 1], Xi, a, false);
  gn_row(P + 16 * i, oxy[2 * i], oxy[2 * i + 1], Xi, b, true);
  const double va[8] = {a.j00, a.j01, a.j02, a.j10, a.j11, a.j12, a.r0, a.r1};
  const double vb[8] = {b.j00, b.j01, b.j02, b.j10, b.j11, b.j12, b.r0, b.r1};
  for (int k = 0; k < 8; k++) {
    out[(size_t)k * n + i] = va[k];
    out[(size_t)(8 + k) * n + i] = vb[k];
  }
}
extern "C" int eg3d_probe_gn_div(uint64_t n, const double* num, const double* den, double* out3n) {
  double *dn, *dd, *dout;
  PT(hipMalloc(&dn, n * 8));
  PT(hipMalloc(&dd, n * 8));
  PT(hipMalloc(&dout, n * 8 * 3));
  PT(hipMemcpy(dn, num, n * 8, hipMemcpyHostToDevice));
  PT(hipMemcpy(dd, den, n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_gn_div, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, dn, dd, dout);
  PT(hipDeviceSynchronize());
  PT(hipMemcpy(out3n, dout, n * 8 * 3, hipMemcpyDeviceToHost));
  (void)hipFree(dn);
  (void)hipFree(dd);
  (void)hipFree(dout);
  return 0;
}
extern "C" int eg3d_probe_gn_rows(uint64_t n, const float* P16, const float* oxy, const double* X, double* out16n) {
  float *dP, *dxy;
  double *dX, *dout;
  PT(hipMalloc(&dP, n * 64));
  PT(hipMalloc(&dxy, n * 8));
  PT(hipMalloc(&dX, n * 24));
  PT(hipMalloc(&dout, n * 8 * 16));
  PT(hipMemcpy(dP, P16, n * 64, hipMemcpyHostToDevice));
  PT(hipMemcpy(dxy, oxy, n * 8, hipMemcpyHostToDevice));
  PT(hipMemcpy(dX, X, n * 24, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_gn_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, dP, dxy, dX, dout);
  PT(hipDeviceSynchronize());
  PT(hipMemcpy(out16n, dout, n * 8 * 16, hipMemcpyDeviceToHost));
  (void)hipFree(dP);
  (void)hipFree(dxy);
  (void)hipFree(dX);
  (void)hipFree(dout);
  return 0;
}

// ---- the lane-group Gauss-Newton solver AT FULL DENSITY (tools/gn_floor.py): every single-wave block solves `rounds` windows
// of SEVEN identical ADD requests of n rows each (7 x 9 = 63 of the 64 rows of a packed round; identical requests converge in
// the same iteration, so no lane waits for another request) through coop_gn_groups — the product's solver, the product's
// build switches, 4 waves per SIMD like the expand kernel. What comes out is the solver's speed of light on this chip under
// the arithmetic contract (FP64, no FMA, ordered sums): row-iterations per second.
__global__ void __launch_bounds__(64, 4) k_probe_gn_dense(const float* cam_P, const Obs* obs, int n, float x0, float y0, float z0,
                                                          int rounds, float* out) {
  __shared__ CoopLds L;
  const int lane = (int)threadIdx.x;
  if (lane == 0) {
    L.cams_mid_range = 1;
    L.long_refused = 0;
  }
  __syncthreads();
  float acc = 0.0f;
  int n_ok = 0;
  for (int r = 0; r < rounds; r++) {
    const float X0[3] = {x0, y0, z0};
    float X[3] = {0.0f, 0.0f, 0.0f};
    const Obs ex = obs[n - 1];
    const bool ok = coop_gn_groups<0, false>(cam_P, L, lane < 7, obs, n - 1, true, (int32_t)ex.view, ex.x, ex.y, X0, X);
    acc += X[0];
    n_ok += ok ? 1 : 0;
  }
  if (lane == 0) {
    out[4 * blockIdx.x] = acc / (float)rounds;
    out[4 * blockIdx.x + 1] = (float)n_ok;
  }
}
// obs_view / obs_xy: the n observations of the request (the last one is the ADD observation); returns the kernel time in ms
// and, in X_ok[0..1], block 0's mean solution x and the number of accepted solves of its lane 0
extern "C" int eg3d_probe_gn_dense(const float* cam_P, int n_views, const int32_t* obs_view, const float* obs_xy, int n,
                                   const float* X0, int n_blocks, int rounds, float* ms, float* X_ok) {
  if (n < 3 || n > 9 || n_blocks < 1 || rounds < 1) return -1;
  float* dP;
  Obs* dobs;
  float* dout;
  std::vector<Obs> h((size_t)n);
  for (int i = 0; i < n; i++) {
    h[i].view = (uint32_t)obs_view[i];
    h[i].pl = 0;
    h[i].seg = 0;
    h[i].x = obs_xy[2 * i];
    h[i].y = obs_xy[2 * i + 1];
  }
  PT(hipMalloc(&dP, (size_t)n_views * 64));
  PT(hipMalloc(&dobs, sizeof(Obs) * (size_t)n));
  PT(hipMalloc(&dout, sizeof(float) * 4 * (size_t)n_blocks));
  PT(hipMemcpy(dP, cam_P, (size_t)n_views * 64, hipMemcpyHostToDevice));
  PT(hipMemcpy(dobs, h.data(), sizeof(Obs) * (size_t)n, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  PT(hipEventCreate(&a));
  PT(hipEventCreate(&b));
  hipLaunchKernelGGL(k_probe_gn_dense, dim3((unsigned)n_blocks), dim3(64), 0, 0, dP, dobs, n, X0[0], X0[1], X0[2], 2, dout);  // warm-up
  PT(hipEventRecord(a, 0));
  hipLaunchKernelGGL(k_probe_gn_dense, dim3((unsigned)n_blocks), dim3(64), 0, 0, dP, dobs, n, X0[0], X0[1], X0[2], rounds, dout);
  PT(hipEventRecord(b, 0));
  PT(hipDeviceSynchronize());
  PT(hipEventElapsedTime(ms, a, b));
  float r[4];
  PT(hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost));
  X_ok[0] = r[0];
  X_ok[1] = r[1];
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(dP);
  (void)hipFree(dobs);
  (void)hipFree(dout);
  return 0;
}

// ---- the 2-D geometry primitives the glm pin test checks (tests/test_glm_pin.py): project_f32, seg_line_cos,
// seg_closest of the product's headers on the GPU. in = [n][19] / [n][7] / [n][6] floats as tests/glm/glm_driver.cpp takes.
__global__ void k_probe_geom(uint64_t n, int mode, const float* in, float* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) {
    const float* q = in + 19 * i;
    float u, v;
    project_f32(q, q[16], q[17], q[18], u, v);
    out[2 * i] = u;
    out[2 * i + 1] = v;
  } else if (mode == 1) {
    const float* q = in + 7 * i;
    out[i] = seg_line_cos(q[0], q[1], q[2], q[3], q[4], q[5]);
  } else {
    const float* q = in + 6 * i;
    float qx, qy;
    out[3 * i] = seg_closest(q[0], q[1], q[2], q[3], q[4], q[5], qx, qy);
    out[3 * i + 1] = qx;
    out[3 * i + 2] = qy;
  }
}
extern "C" int eg3d_probe_geom(uint64_t n, int mode, const float* in, float* out) {
  const size_t rin = mode == 0 ? 19 : mode == 1 ? 7 : 6, rout = mode == 0 ? 2 : mode == 1 ? 1 : 3;
  float *di, *dout;
  PT(hipMalloc(&di, n * rin * 4));
  PT(hipMalloc(&dout, n * rout * 4));
  PT(hipMemcpy(di, in, n * rin * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_probe_geom, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, mode, di, dout);
  PT(hipDeviceSynchronize());
  PT(hipMemcpy(out, dout, n * rout * 4, hipMemcpyDeviceToHost));
  (void)hipFree(di);
  (void)hipFree(dout);
  return 0;
}
