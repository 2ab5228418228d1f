/*
 * eg3d_probe.h — TEST-ONLY entry points of tests/probe/libeg3d_probe.so: single device primitives of
 * the product's headers run on the GPU, for the bit-for-bit arithmetic checks of
 * tests/test_gpu_arith.py. Not part of the product and not linked into libeg3d.so.
 */
#ifndef EG3D_PROBE_H_
#define EG3D_PROBE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* out_d[5][n]: a/b, sqrt(|a|), a*b+c (two roundings), (double)(float)a, 1/sqrt; out_f likewise in float */
int eg3d_probe_arith(uint64_t n, const double* a, const double* b, const double* c, double* out_d, const float* fa,
                     const float* fb, const float* fc, float* out_f);
/* n_cases triangulations of k observations each; cam_P = [n_views][16] host array */
int eg3d_probe_dlt_rows(void); /* the DLT form the probe was compiled with (EG3D_DLT_ROWS) */
int eg3d_probe_triangulate(const float* cam_P, int n_views, uint64_t n_cases, int k, const int32_t* views, const float* xy,
                           float* X, uint8_t* valid, double* dlt_X0);
/* the same 2-view DLTs (minimum view id, last observation) on groups of 8 lanes (dlt2_grp8, eg3d_dev_coopgn.h) */
int eg3d_probe_dlt_groups(const float* cam_P, int n_views, uint64_t n_cases, int k, const int32_t* views, const float* xy,
                          double* dlt_X0);
/* the shared-reciprocal division of the Gauss-Newton rows: out[0..n) = num/den, out[n..2n) = gn_div(num, gn_recip(den)),
 * out[2n..3n) = 1.0 where den is in the admitted range */
int eg3d_probe_gn_div(uint64_t n, const double* num, const double* den, double* out3n);
/* GnRow (j00 j01 j02 j10 j11 j12 r0 r1) of n rows with the plain divisions (out[0..8n)) and through the guarded fast
 * path (out[8n..16n)); P16 = one 4x4 camera matrix per row */
int eg3d_probe_gn_rows(uint64_t n, const float* P16, const float* oxy, const double* X, double* out16n);
/* the lane-group Gauss-Newton solver at full density (tools/gn_floor.py): n_blocks single-wave blocks x rounds windows of
 * seven identical ADD requests of n <= 9 rows; *ms = kernel time; X_ok = {mean solution x of block 0, its accepted solves} */
int eg3d_probe_gn_dense(const float* cam_P, int n_views, const int32_t* obs_view, const float* obs_xy, int n,
                        const float* X0, int n_blocks, int rounds, float* ms, float* X_ok);
/* 2-D geometry primitives on the GPU (tests/test_glm_pin.py): mode 0 project_f32 (in [n][19] = P16 + X -> [n][2]),
 * 1 seg_line_cos (in [n][7] = segment + line -> [n]), 2 seg_closest (in [n][6] = p, v, w -> [n][3] = d2, closest point) */
int eg3d_probe_geom(uint64_t n, int mode, const float* in, float* out);
#ifdef __cplusplus
}
#endif
#endif
