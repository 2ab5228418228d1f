/*
 * eg3d_host.h — host-side (no GPU) utilities of the MI355X-native EdgeGraph3D path:
 * the seeded synthetic workload generator of SURVEY.md 8(d), the grid-map builder used by
 * eg3d_create (row a3), the OpenMVG-JSON reader/writer (row a-IO), and the host steps
 * either side of the path (N3: dedup + observation filter). Plain C ABI, POD only.
 */
#ifndef EG3D_HOST_H_
#define EG3D_HOST_H_
#include "eg3d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------ synthetic workload ---- */
typedef struct eg3d_synth_config {
  int32_t n_views;
  uint32_t n_seeds;
  int32_t n_curves;       /* 3-D curves in the 400 mm cube */
  uint64_t rng_seed;      /* master seed; SURVEY 8(d): 0xE63D2018 + config index stream */
  int32_t max_track;      /* k ~ U[3, min(V, max_track)] */
  float obs_noise_px;     /* seed observation noise sigma (0.4) */
  float vtx_noise_px;     /* polyline vertex noise sigma (0.15) */
  float invalid_frac;     /* fraction of polylines emitted invalid/empty (0.01) */
  float seed_offset_px;   /* seeds lie within this many projected px of a curve (6) */
  int32_t width, height;  /* 1600 x 1200 */
  float focal, ppx, ppy;  /* 2890, 823, 619 */
} eg3d_synth_config;

typedef struct eg3d_synth eg3d_synth;

void eg3d_synth_default_config(eg3d_synth_config* c, int config_index /* 2,3,4: SURVEY C2,C3',C4; 0: tiny */);
eg3d_synth* eg3d_synth_create(const eg3d_synth_config* c);
const eg3d_scene* eg3d_synth_scene(const eg3d_synth* s);
const eg3d_seeds* eg3d_synth_seeds(const eg3d_synth* s);
/* true 3-D position of each seed's curve point ([n_seeds][3]) — for sanity checks only */
const float* eg3d_synth_seed_truth(const eg3d_synth* s);
uint64_t eg3d_synth_total_segments(const eg3d_synth* s);
/* index of the 3-D curve every polyline was generated from ([view_pl_off[V]], global polyline
 * order) — ground truth for synthetic "potentially compatible polylines" sets (pipelines 1-2) */
const uint32_t* eg3d_synth_polyline_curve(const eg3d_synth* s);
int eg3d_synth_n_curves(const eg3d_synth* s);
void eg3d_synth_destroy(eg3d_synth* s);

/* Config 5 workload: n points with k~U[3,10] observations, X = truth + N(0, 2 mm),
 * obs = projection + N(0, 0.5 px), 5 % gross outliers (20-50 px). Arrays are malloc'd;
 * free with eg3d_host_free. */
int eg3d_synth_points(const eg3d_synth* s, uint64_t n_points, uint64_t rng_seed, float** X, uint32_t** obs_off,
                      int32_t** obs_view, float** obs_xy);
void eg3d_host_free(void* p);

/* camera of the synthetic rig (for writing the scene as an OpenMVG file): K = (focal, ppx, ppy),
 * R row-major, C = centre */
int eg3d_synth_camera(const eg3d_synth* s, int view, float* focal, float* ppx, float* ppy, float* R9, float* C3);

/* ------------------------------------------------------ polyline-graph file ---- */
/* The reference builds its PolyLineGraph2DHMapImpl per view from the edge images at start-up
 * (io/input/convert_edge_images_pixel_to_segment.cpp:868-892, SURVEY N2: out of scope); this
 * container ("EG3DPLG1", layout in edgegraph3d_amd/host/plg_file.cpp) carries the polyline graphs
 * of all views — ids = the reference's vector positions — from whatever built them to the path. */
typedef struct eg3d_plg eg3d_plg;
int eg3d_plg_write(const char* path, const eg3d_scene* scene);
eg3d_plg* eg3d_plg_read(const char* path);
/* polyline part of an eg3d_scene (cam_P / F / F_valid null: taken from the SfM data by the caller) */
const eg3d_scene* eg3d_plg_scene(const eg3d_plg* g);
void eg3d_plg_destroy(eg3d_plg* g);

/* ---------------------------------------- edge image -> polyline graph (N2) ---- */
/* convertEdgeImagePolyLineGraph_optimized (io/input/convert_edge_images_pixel_to_segment.cpp:294-426,
 * 428-626, 868-883) + PolyLineGraph2DHMapImpl::optimize (plgs/polyline_graph_2d_hmap_impl.cpp:255-266):
 * a binary edge image becomes a pixel graph (8-neighbourhood, short cycles avoided), the graph a
 * polyline graph, which is then simplified (1 px), its 2-connection nodes merged, close extremes
 * connected and weak components dropped. Polyline and node ids = the reference's vector positions
 * (invalidated polylines keep their id, with no vertices). One view per call; independent views
 * may be built concurrently. */
typedef struct eg3d_plg_view {
  uint32_t n_polylines;
  uint32_t* pl_vtx_off;   /* [n_polylines+1] */
  float* vtx_xy;          /* [pl_vtx_off[n_polylines]][2] */
  uint32_t* pl_start;     /* [n_polylines] node ids */
  uint32_t* pl_end;
  uint8_t* pl_valid;      /* PolyLineGraph2D::is_valid_polyline */
  uint32_t n_nodes;
  float* node_xy;         /* [n_nodes][2] nodes_coords ((-1,-1) = invalidated) */
} eg3d_plg_view;
/* mask: height*width bytes, non-zero = edge colour (EDGE_COLOR 255,255,255 as cv::imread(IMREAD_COLOR) sees it) */
int eg3d_plg_build_from_mask(const uint8_t* mask, int width, int height, eg3d_plg_view* out);
/* PNG reader of the edge images (8/16-bit or 1/2/4-bit grey, RGB, palette, with or without alpha,
 * non-interlaced): mask[i] = 1 where the pixel is white. *mask is malloc'd (eg3d_host_free). */
int eg3d_png_read_edge_mask(const char* path, int* width, int* height, uint8_t** mask);
int eg3d_plg_build_from_png(const char* path, int* width, int* height, eg3d_plg_view* out);
void eg3d_plg_view_free(eg3d_plg_view* v);
/* All views of a scene, built concurrently on the host's cores (the reference builds them one after the other:
 * convert_edge_images_pixel_to_segment.cpp:868-892). out[n_views]; 0, or -(v + 1) for the first view whose image cannot be
 * read or differs in size from view 0's (nothing is left allocated then). */
int eg3d_plg_build_views_from_png(const char* const* paths, int n_views, int* width, int* height, eg3d_plg_view* out);
/* assembles per-view graphs into the container the path consumes (eg3d_plg_scene) */
eg3d_plg* eg3d_plg_from_views(int n_views, int width, int height, const eg3d_plg_view* views);

/* --------------------------------------------------------------- grid maps ---- */
/* Uniform grid of one view (reference PolyLine2DMap ctor, polyLine_2d_map.cpp:40-58).
 * Outputs malloc'd CSR arrays (cell = row*ncols+col), free with eg3d_host_free. */
int eg3d_host_build_grid(const eg3d_scene* scene, int view, float cell_dim, uint32_t* ncols, uint32_t* nrows,
                         uint32_t** cell_off, uint32_t** ids, uint32_t* dropped);

/* ------------------------------------------------------------ host post steps -- */
/* filter_3d_points_close_2d_array (filtering_close_plgps.cpp:99-124): keep[i] = 1 if kept. */
int eg3d_host_filter_close_2d(int n_views, int width, int height, const eg3d_edgepoints* pts, uint8_t* keep);
/* compute_inliers tail (outliers_filtering.cpp:37-64); returns the threshold used. */
int eg3d_host_observation_filter(int n_cameras, const uint32_t* obs_off, uint64_t n_points, uint64_t first_edgepoint,
                                 int forced_min_filter, uint8_t* inlier_inout);

/* ------------------------------------------ the cloud exchange on host arrays -- */
/* The arithmetic of eg3d_allgather_edgepoints (include/eg3d_rccl.h) for clouds held in HOST memory — what a
 * CPU / non-RCCL transport (MPI, gloo) needs around its own all-gather of the raw arrays, and what the
 * world_size-2 gloo test of the multi-GPU path drives. counts3[r] = {n_points, n_obs, status} of rank r.
 * eg3d_host_gather_plan fills the per-rank bases (where rank r's points / observations start in the gathered
 * cloud) and the totals; returns 0, or -4 (EG3D_GATHER_ERR_INCOMPLETE) when any status is non-zero.
 * eg3d_host_gather_place copies rank r's cloud `part` to its place in `whole` (arrays allocated by the caller for
 * the totals; obs_off has total_points + 1 entries) and rebases its observation offsets by obs_base;
 * the caller writes the sentinel whole->obs_off[total_points] = total_obs. */
int eg3d_host_gather_plan(int n_ranks, const uint64_t* counts3, uint64_t* point_base, uint64_t* obs_base,
                          uint64_t* total_points, uint64_t* total_obs);
int eg3d_host_gather_place(const eg3d_edgepoints* part, uint64_t point_base, uint64_t obs_base, eg3d_edgepoints* whole);

/* ------------------------------------------- PLGMatchesManager replay (row a17) -- */
/* What plgmm.add_matched_3dpolyline(chain) leaves behind when the reference runs the path
 * (plg_matching_from_refpoints.cpp:74-77 -> plg_matches_manager.cpp:99-180): the 3-D polyline graph
 * (PolyLineGraph3DHMapImpl: nodes keyed by exact coordinates, one 2-point polyline per distinct
 * consecutive pair of chain points, polyline_graph_3d_hmap_impl.cpp:47-68,99-141) and the matched
 * 2-D intervals per (view, polyline). Rebuilt on the host from the ordered edge-point cloud of
 * eg3d_match_refpoints / eg3d_match_polyline_sets (chains = runs of equal key[0..2] with key[3]
 * counting up). Node and polyline ids are the reference's (creation order), narrowed to 32 bits. */
typedef struct eg3d_graph3d {
  uint64_t n_nodes;          /* nodes_amount, invalidated nodes included */
  uint64_t n_real_nodes;     /* real_nodes_amount */
  float* node_X;             /* [n_nodes][3] nodes_coords; an invalidated node holds (-1,-1,-1) */
  uint64_t* node_point;      /* [n_nodes] edge-point whose observations the node carries (set_observations /
                                p3d_to_matches_map: last writer), ~0 if none */
  uint64_t n_polylines;      /* 2-point polylines = direct connections, duplicates suppressed */
  uint32_t* pl_start;        /* [n_polylines] node ids */
  uint32_t* pl_end;
  uint64_t* conn_off;        /* [n_nodes+1] connections[node]: polyline ids in insertion order */
  uint32_t* conn_pl;
  uint64_t n_scene_polylines;/* = view_pl_off[V] */
  uint64_t* iv_off;          /* [n_scene_polylines+1] matched_polyline_intervals[view][polyline] as a CSR over
                                the scene's global polyline index, ascending start segment */
  uint32_t* iv_start_seg;
  float* iv_start_xy;        /* [..][2] */
  uint32_t* iv_end_seg;
  float* iv_end_xy;
} eg3d_graph3d;
int eg3d_host_replay_matches(const eg3d_scene* scene, const eg3d_edgepoints* pts, eg3d_graph3d* out);
void eg3d_host_free_graph3d(eg3d_graph3d* g);

/* -------------------------------------------------------------- OpenMVG JSON --- */
typedef struct eg3d_sfm eg3d_sfm; /* host mirror of SfMData (SfMData.h:16-30) */
eg3d_sfm* eg3d_sfm_read_json(const char* path);          /* OpenMvgParser::parse, OpenMvgParser.cpp:39-301 */
int eg3d_sfm_write_json(const eg3d_sfm* s, const char* in_path_for_passthrough, const char* out_path); /* output_sfm_data.cpp:186-229 */
/* The writer's text rules on their own (json_text.hpp): a finite double as rapidjson's Writer prints it (Grisu2 +
 * notation); a JSON number literal as the reference's reader (default flags) + writer re-print it; the computed
 * Grisu cached power 10^(-348 + 8*index), index 0..86. Return the text length, or -1. Checked against the reference
 * tree's vendored rapidjson by tests/test_json_rapidjson.py. */
/* The camera model eg3d_sfm_read_json / eg3d_sfm_set_camera apply (OpenMvgParser.cpp:289 t = -center * rotation,
 * :107-125 cameraMatrix = eMatrix * kMatrix, in glm's float evaluation order), for n cameras on bare arrays:
 * fpp = [n][3] focal, ppx, ppy; R9 = [n][9] row-major rotation; C3 = [n][3] centre -> t3 [n][3], P16 [n][16]. */
int eg3d_host_camera_model(uint64_t n, const float* fpp, const float* R9, const float* C3, float* t3, float* P16);
int eg3d_host_json_double_text(double d, char* buf, int cap);
int eg3d_host_json_number_text(const char* literal, char* buf, int cap);
int eg3d_host_json_cached_power(int index, uint64_t* f, int* e);
eg3d_sfm* eg3d_sfm_create(int n_views, int width, int height);
void eg3d_sfm_destroy(eg3d_sfm* s);
int eg3d_sfm_n_views(const eg3d_sfm* s);
uint64_t eg3d_sfm_n_points(const eg3d_sfm* s);
const float* eg3d_sfm_cam_P(const eg3d_sfm* s);           /* [V][16] */
const char* eg3d_sfm_image_path(const eg3d_sfm* s, int view);   /* SfMData::camerasPaths_[view] (valid until the handle is destroyed) */
int eg3d_sfm_image_size(const eg3d_sfm* s, int* width, int* height); /* SfMData::imageWidth_ / imageHeight_ */
int eg3d_sfm_set_camera(eg3d_sfm* s, int view, float focal, float ppx, float ppy, const float* R9_rowmajor,
                        const float* center3, const char* image_path);
/* seeds view of the points (pointers valid until the next mutation) */
int eg3d_sfm_seeds(const eg3d_sfm* s, eg3d_seeds* out);
const float* eg3d_sfm_points(const eg3d_sfm* s);          /* [N][3] */
int eg3d_sfm_add_point(eg3d_sfm* s, const float* X3, int n_obs, const int32_t* views, const float* xy);
/* add_3dpoints_to_sfmd (output_utilities.cpp:96-111) */
int eg3d_sfm_add_edgepoints(eg3d_sfm* s, const eg3d_edgepoints* pts, const uint8_t* keep /* may be NULL */);
/* removeOutliers (outliers_filtering.cpp:66-92) */
int eg3d_sfm_remove_outliers(eg3d_sfm* s, const uint8_t* inlier);
int eg3d_sfm_set_point_coords(eg3d_sfm* s, const float* X /* [N][3] */);
/* analytic fundamental matrices from the cameras (replaces generate_all_fundamental_matrices,
 * geometric_utilities.cpp:818-820, whose LMedS estimate is not reproducible without OpenCV) */
int eg3d_sfm_analytic_F(const eg3d_sfm* s, double* F /* [V][V][9] */, uint8_t* F_valid /* [V][V] */);

/* ------------------------------------ fundamental matrices from the tracks (row N4) -- */
/* generate_all_fundamental_matrices_from_Points / findFundamentalMatrixFromPoints
 * (geometric_utilities.cpp:754-820): for every ORDERED pair of views the points seen from both
 * (ascending id; a point's position on a view = its LAST observation with that view id); pairs with
 * fewer than 10 common points get no matrix (F_valid = 0: every epipolar line of the pair fails, as
 * with the reference's 1x1 Mat). n_common (may be NULL) receives the counts.
 *   estimate = 0: only F_valid / n_common are written — the reference's validity rule, to be combined
 *                 with eg3d_sfm_analytic_F (F may be NULL);
 *   estimate = 1: F[i][j] (l_j = F x_i) is estimated from the correspondences by the build's own
 *                 least-median-of-squares estimator (normalised 8-point samples, deterministic in
 *                 rng_seed). The reference uses cv::findFundamentalMat(FM_LMEDS), a randomised OpenCV
 *                 routine: the matrices are NOT comparable bit for bit, only geometrically.
 * Returns the number of pairs whose estimate failed (>= 0), < 0 on bad arguments. */
int eg3d_host_estimate_F(int n_views, uint64_t n_points, const uint32_t* trk_off, const int32_t* trk_view,
                         const float* trk_xy, int estimate, uint64_t rng_seed, double* F /* [V][V][9] */,
                         uint8_t* F_valid /* [V][V] */, uint32_t* n_common /* [V][V] or NULL */);
int eg3d_sfm_estimate_F(const eg3d_sfm* s, int estimate, uint64_t rng_seed, double* F, uint8_t* F_valid,
                        uint32_t* n_common);

#ifdef __cplusplus
}
#endif
#endif
