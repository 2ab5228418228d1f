/*
 * eg3d_probe.h — diagnostic entry points of libeg3d.so: they run single device primitives
 * (IEEE arithmetic, triangulation, polyline walks) on the GPU so the tests can check the
 * arithmetic contract bit-for-bit against the CPU oracle. Not part of the drop-in surface.
 */
#ifndef EG3D_PROBE_H_
#define EG3D_PROBE_H_
#include "eg3d.h"
#ifdef __cplusplus
extern "C" {
#endif
/* out_d[5][n]: a/b, sqrt(|a|), a*b+c (two roundings), (double)(float)a, 1/sqrt; out_f likewise in float */
int eg3d_probe_arith(eg3d_ctx* ctx, uint64_t n, const double* a, const double* b, const double* c, double* out_d,
                     const float* fa, const float* fb, const float* fc, float* out_f);
/* n_cases triangulations of k observations each (views index ctx's cameras) */
int eg3d_probe_triangulate(eg3d_ctx* ctx, uint64_t n_cases, int k, const int32_t* views, const float* xy, float* X,
                           uint8_t* valid, double* dlt_X0);
/* Per-section shader-clock ticks of the most recent k3b_expand launch, summed over chains
 * (sum[12]) and of the slowest chain (slowest[12]); all zero unless the library was built with
 * -DEG3D_SECTION_TIMING. Index: 0 candidates, 1 N-view step walks, 2 side walks, 3 batched GN,
 * 4 chain following (includes 1, 5, 6, 11), 5 step DLT, 6 step GN, 7 whole chain, 8 commit,
 * 9 chain init, 10 epipolar-candidate pre-solves, 11 new point. */
int eg3d_probe_sections(eg3d_ctx* ctx, double* sum, double* slowest, uint32_t* n_chains);
/* Same for k3a_hypotheses (per hypothesis): 0 first TRI, 1 orient, 2 replay, 3 opposite test,
 * 4 follow dir1, 5 follow dir2; sum over hypotheses and the slowest hypothesis; n by status. */
int eg3d_probe_hyp_sections(eg3d_ctx* ctx, double* sum, double* slowest, uint32_t* counts /*[5]: n, tri, d1, d2, compat*/);
#ifdef __cplusplus
}
#endif
#endif
