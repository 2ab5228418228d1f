/* ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (no reference tests / golden vectors
 * exist, and the reference cannot be built here — SURVEY.md F3/F4).
 * C interface of the CPU restatement, consumed through ctypes by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg. Never linked into the product. */
#ifndef EG3D_ORACLE_H_
#define EG3D_ORACLE_H_
#include "../include/eg3d_host.h" /* eg3d.h + the eg3d_graph3d layout */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

typedef struct orc_stats {
  uint64_t n_tasks, n_hyp, n_chains;
  uint64_t n_tri, n_add, n_degenerate_dlt, n_combos;
  uint64_t bytes_algorithmic;
  uint32_t dir_mismatch;
  uint32_t grid_dropped;
  double seconds; /* wall time of the call, scene/grid construction excluded */
} orc_stats;

/* which system cv::triangulatePoints builds (process-wide): 2 rows per view = 4x4 (later OpenCV), 3 = 6x4
 * (OpenCV 2.4-3.1, incl. the x*P1 - y*P0 row); must match the product's eg3d_dlt_rows() */
int orc_set_dlt_rows(int rows);
int orc_get_dlt_rows(void);
/* test hook: bit q = behave as a "corrected" implementation of quirk Qq would (q = 4, 12, 13); 0 = reference */
void orc_set_quirk_fixes(unsigned mask);
/* TEST HOOK: arithmetic conventions of the unpinned OpenCV routines (oracle_tri.hpp g_conv); 0 = the restatement */
void orc_set_conventions(unsigned mask);
orc_ctx* orc_create(const eg3d_scene* scene);
void orc_destroy(orc_ctx*);
int orc_get_grid(orc_ctx*, int view, int which, uint32_t* ncols, uint32_t* nrows, const uint32_t** cell_off,
                 const uint32_t** ids);
/* out uses the eg3d_edgepoints layout; release with orc_free_edgepoints. nthreads<=1: serial. */
int orc_match_refpoints(orc_ctx*, const eg3d_seeds* seeds, uint32_t seed_begin, uint32_t seed_end, int nthreads,
                        eg3d_edgepoints* out, orc_stats* stats);
/* Pipelines 1-2 extractor: find_new_3d_points_from_compatible_polylines_expandallviews_parallel
 * (polyline_matching.cpp:153-208) over sets [set_begin, set_end). Sets are a CSR over rows
 * (set * n_views + view): row_off[n_sets*n_views + 1], pl_ids ascending per row. key = (sample of
 * the call, start view, 0, index in chain). */
int orc_match_polyline_sets(orc_ctx*, uint32_t n_sets, const uint32_t* row_off, const uint32_t* pl_ids,
                            uint32_t set_begin, uint32_t set_end, int nthreads, eg3d_edgepoints* out, orc_stats* stats);
void orc_free_edgepoints(eg3d_edgepoints* e);
int orc_candidates(orc_ctx*, const eg3d_seeds* seeds, uint32_t seed_begin, uint32_t seed_end, eg3d_candidates* out);
void orc_free_candidates(eg3d_candidates* c);
int orc_gn_filter(orc_ctx*, const float* X, const uint32_t* obs_off, const int32_t* obs_view, const float* obs_xy,
                  uint64_t n_points, float gn_max_mse, int legacy_abs, int nthreads, float* X_out, uint8_t* inlier);
/* filter_3d_points_close_2d_array (filtering_close_plgps.cpp:99-124): keep[i]=1 for kept points */
int orc_filter_close_2d(orc_ctx*, const eg3d_edgepoints* pts, uint8_t* keep);
/* compute_inliers tail (outliers_filtering.cpp:37-64) given GN inliers: returns threshold used */
int orc_observation_filter(int n_cameras, const uint32_t* obs_off, uint64_t n_points, uint64_t first_edgepoint,
                           int forced_min_filter, uint8_t* inlier_inout);

/* row a17: PLGMatchesManager::add_matched_3dpolyline replayed over the chains of a cloud
 * (plg_matches_manager.cpp:99-180); release with orc_free_graph3d */
int orc_replay_matches(orc_ctx*, const eg3d_edgepoints* pts, eg3d_graph3d* out);
void orc_free_graph3d(eg3d_graph3d* g);

/* binary edge image -> optimised polyline graph (convert_edge_images_pixel_to_segment.cpp:879-883);
 * mask = height*width bytes, non-zero = edge; release with orc_free_plg_view */
int orc_plg_from_mask(const uint8_t* mask, int width, int height, eg3d_plg_view* out);
void orc_free_plg_view(eg3d_plg_view* v);

/* primitive probes for known-answer tests */
/* N4, the deterministic part of findFundamentalMatrixFromPoints (geometric_utilities.cpp:754-779): the common
 * points of the ordered view pair (i, j) and their positions, or n = 0 when there are fewer than 10.
 * ids / xy_i / xy_j must hold n_points entries; returns n. */
int orc_pair_correspondences(int n_views, uint64_t n_points, const uint32_t* trk_off, const int32_t* trk_view,
                             const float* trk_xy, int i, int j, uint32_t* ids, float* xy_i, float* xy_j,
                             uint32_t* n_common);
float orc_squared_2d_distance(float ax, float ay, float bx, float by);
float orc_minimum_distancesq(float px, float py, float vx, float vy, float wx, float wy, float* proj);
int orc_intersect_segment_line(float x1, float y1, float x2, float y2, const float* line, float* inter,
                               int* parallel, int* overlapped);
int orc_intersect_segment_line_nqp(float x1, float y1, float x2, float y2, const float* line, float* inter,
                                   int* quasiparallel, float* distance);
int orc_cell_from_coords(float cell, float x, float y, int* col, int* row, int* b_row, int* b_col);
int orc_epiline(const double* F9, float x, float y, float* line);
void orc_project(const float* P16, const float* X, float* xy);
/* batched probes (tests/test_glm_pin.py: the oracle's evaluation orders against the reference's vendored glm) */
void orc_batch_project(uint64_t n, const float* P16 /*[n][16]*/, const float* X /*[n][3]*/, float* xy /*[n][2]*/);
void orc_batch_mindist(uint64_t n, const float* pvw /*[n][6]*/, float* out /*[n][3]: d2, projection*/);
void orc_batch_anglecos(uint64_t n, const float* seg_line /*[n][7]: x1 y1 x2 y2 a b c*/, float* out /*[n]*/);
int orc_triangulate(const float* P /*[n][16]*/, const int* view_ids, const float* xy, int n, float* X, int* degenerate);
int orc_gn_add(const float* P, const int* view_ids, const float* xy, int n, const float* X0, float* X);
void orc_dlt(const float* P1, const float* xy1, const float* P2, const float* xy2, double* X0);
/* polyline walking probes: vtx = [n][2], start/end node ids */
int orc_next_by_distance(const float* vtx, int n, uint32_t start, uint32_t end, uint32_t seg, float x, float y,
                         uint32_t direction, float distance, uint32_t* out_seg, float* out_xy);
int orc_next_by_line(const float* vtx, int n, uint32_t start, uint32_t end, uint32_t seg, float x, float y,
                     uint32_t direction, const float* line, int bounded, float mind, float maxd, uint32_t* out_seg,
                     float* out_xy, int* flags /*1 qp,2 extreme,4 bound*/);

#ifdef __cplusplus
}
#endif
#endif
