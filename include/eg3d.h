/*
 * eg3d.h — C ABI of the MI355X-native EdgeGraph3D hot path
 * (refpoint -> epipolar polyline match -> multi-view edge-point triangulation,
 *  plus the batched Gauss-Newton outlier filter).
 *
 * The reference (abignoli/EdgeGraph3D) has no FFI layer: its seams are C++ free
 * functions and virtual classes. Each entry point below names the reference seam
 * it replaces (paths relative to the reference tree):
 *
 *   eg3d_create / eg3d_clone / eg3d_destroy
 *       context construction in edge_matching(): PolyLine2DMapSearch per view at
 *       4 px (src/edgegraph3d/edge_matcher.cpp:101-103), PLGEdgeManager ctor with
 *       the 30 px maps (edge_managers/plg_edge_manager.cpp:46-75),
 *       PLGPCM3ViewsPLGFollowing ctor (edge_matcher.cpp:115).
 *   eg3d_candidates
 *       PLGEdgeManager::detect_nearby_intersections_and_correspondences_plgp(int)
 *       (include/.../plg_edge_manager.hpp:74, plg_edge_manager.cpp:261-300).
 *   eg3d_match_refpoints
 *       plg_matching_from_refpoints[_parallel](sfmd, em, cm, plgmm)
 *       (include/.../plg_matching_from_refpoints.hpp:53,55;
 *        src/.../plg_matching_from_refpoints.cpp:64-116).
 *   eg3d_gn_filter
 *       gaussNewtonFiltering(SfMData&, vector<bool>&, float)
 *       (include/edgegraph3d/filtering/gauss_newton.hpp:20; gauss_newton.cpp:136-178).
 *   eg3d_get_grid
 *       read-back of PolyLine2DMap::pls_id_maps (matching/plg_matching/polyLine_2d_map.cpp:40-58)
 *       so the grid membership can be checked against the CPU oracle.
 *
 * Conventions: plain C, POD structs, caller-owned input buffers that must stay
 * valid for the duration of the call only (eg3d_create copies the scene to HBM).
 * Status: 0 = ok, <0 = error (eg3d_last_error() gives text). One context per GPU;
 * a context is thread-compatible (one caller thread at a time), like the reference.
 * Ids that are `ulong` in the reference are uint32_t here (documented narrowing).
 */
#ifndef EG3D_H_
#define EG3D_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EG3D_OK 0
#define EG3D_ERR_ARG -1
#define EG3D_ERR_HIP -2
#define EG3D_ERR_CAPACITY -3 /* a device-side fixed capacity was exceeded; flags say which */
#define EG3D_ERR_NODEVICE -4

/* ---------------------------------------------------------------- scene ---- */
/* Flat, read-only description of everything the path reads besides the seeds.
 * Replaces SfMData::camerasList_[v].cameraMatrix, `Mat** F` and
 * vector<PolyLineGraph2DHMapImpl> (reference: SfMData.h:16-30,
 * types_reconstructor.hpp:68-82, polyline_graph_2d.hpp:85-119,278-294). */
typedef struct eg3d_scene {
  int32_t n_views;            /* <= 8192; a view holds <= 524 288 polylines (EG3D_ERR_CAPACITY otherwise) */
  int32_t width, height;      /* image size, identical for all views (imgs[0].size()) */
  const float* cam_P;         /* [V][16] cameraMatrix[r][c], row-major 4x4, last row 0 (Q6) */
  const double* F;            /* [V][V][9] row-major; F[i][j] maps a point of view i to its line in view j */
  const uint8_t* F_valid;     /* [V][V] 0 => "1x1 Mat" => epiline fails (geometric_utilities.cpp:826,840) */
  const uint32_t* view_pl_off;/* [V+1] polyline index range of each view in the arrays below */
  const uint32_t* pl_vtx_off; /* [NP+1] vertex range of each polyline (global indices into vtx_xy) */
  const float* vtx_xy;        /* [NV][2] polyline_coords; on valid polylines finite and within +-1e7 px (eg3d_create refuses
                                 the scene otherwise: the reference's grid sampling does not terminate on such input) */
  const uint32_t* pl_start;   /* [NP] node id `start` */
  const uint32_t* pl_end;     /* [NP] node id `end`   */
  const uint8_t* pl_valid;    /* [NP] PolyLineGraph2D::is_valid_polyline (polyline_graph_2d.cpp:1141-1147); the vertices of an
                                 invalid polyline, if the caller left any, are ignored (an invalidated polyline of the
                                 reference has none) */
} eg3d_scene;

/* Seeds = SfM reference points: camViewingPointN_ + point2DoncamViewingPoint_ as CSR. */
typedef struct eg3d_seeds {
  uint32_t n_seeds;
  const uint32_t* trk_off;    /* [N+1] */
  const int32_t* trk_view;    /* [trk_off[N]] view ids, in the order stored in the SfM file */
  const float* trk_xy;        /* [trk_off[N]][2] */
} eg3d_seeds;

/* ------------------------------------------------------------- outputs ----- */
/* Edge-points = vector<tuple<vec3, vector<plg_point>, vector<int>>> of the reference,
 * flattened. Library-owned; release with eg3d_free_edgepoints. Order is the
 * reference's emission order (seed, start view in track order, start hit in
 * ascending polyline id, chain order). */
typedef struct eg3d_edgepoints {
  uint64_t n_points;
  uint64_t n_obs;
  float* X;            /* [n_points][3] */
  uint64_t* obs_off;   /* [n_points+1]; 64-bit: the cloud of one call may hold more than 2^32 observations
                          (BASELINE config 4 as one call: 56 M points, 4.1 G observations) */
  int32_t* obs_view;   /* [n_obs] */
  uint32_t* obs_pl;    /* [n_obs] polyline id inside its view */
  uint32_t* obs_seg;   /* [n_obs] segment index */
  float* obs_xy;       /* [n_obs][2] */
  uint32_t* key;       /* [n_points][4] seed, start-view index in track, start-hit index, index in chain */
  uint64_t n_tasks;        /* (seed, start view, start hit) tasks examined */
  uint64_t n_hypotheses;   /* 3-view hypotheses evaluated */
  uint64_t n_chains;       /* chains emitted */
  uint32_t flags;          /* EG3D_FLAG_* capacity / quirk indicators */
  void* _owner;            /* internal */
} eg3d_edgepoints;

#define EG3D_FLAG_CHAIN_OVERFLOW   1u  /* a chain exceeded the device chain capacity */
#define EG3D_FLAG_OBS_OVERFLOW     2u  /* a chain's observation pool was exhausted   */
#define EG3D_FLAG_HYP_OVERFLOW     4u  /* a following direction exceeded its capacity */
#define EG3D_FLAG_DIR_MISMATCH     8u  /* Q15: a walk was asked to follow a node id that is neither end of its polyline */
#define EG3D_FLAG_DEGENERATE_DLT  16u  /* Q11: a DLT initialisation used the same camera twice */

/* Stage A result for one seed range (testable alone). Library-owned. */
typedef struct eg3d_candidates {
  uint32_t n_sv;            /* number of (seed, track entry) pairs = trk_off[end]-trk_off[begin] */
  uint32_t* cand_off;       /* [n_sv+1] candidate polylines (<= 30 px), ascending id */
  uint32_t* cand_pl;
  uint32_t* start_off;      /* [n_sv+1] starting intersections (<= 10 px) */
  uint32_t* start_pl;
  uint32_t* start_seg;
  float* start_xy;          /* [..][2] */
  /* per starting intersection (task), per track entry of the seed: epipolar hits */
  uint32_t n_tasks;
  uint32_t* task_sv;        /* [n_tasks] index of the (seed,view) pair the task starts from */
  uint32_t* task_hit;       /* [n_tasks] index of the start hit inside that pair */
  uint32_t* task_list_off;  /* [n_tasks+1] offset into list_off: one list per track entry */
  uint32_t* list_off;       /* [n_lists+1] */
  uint32_t* hit_pl;
  uint32_t* hit_seg;
  float* hit_xy;
  void* _owner;
} eg3d_candidates;

typedef struct eg3d_ctx eg3d_ctx;

/* Per-call timing of the device stages, filled by eg3d_match_refpoints (ms, HIP events
 * on the context's stream). */
typedef struct eg3d_stage_times {
  float ms_total;
  float ms_candidates;   /* K1 seed_candidates (count+fill) */
  float ms_epipolar;     /* K2 epipolar_hits (count+fill)   */
  float ms_hypotheses;   /* K3a consensus hypotheses        */
  float ms_select;       /* K3s uniqueness / chain assembly */
  float ms_expand;       /* K3b expand-all-views            */
  float ms_emit;         /* K4 compaction                   */
  uint64_t bytes_algorithmic; /* SURVEY 8(d) algorithmic bytes of this call */
  float ms_slowest_chain; /* the longest time ONE chain held its wavefront in the expand stage: a lower bound of every
                             expand launch, whatever else the GPU is doing (appended in round 6: the struct grew, rebuild callers) */
} eg3d_stage_times;

const char* eg3d_last_error(void);
int eg3d_device_count(void);
/* Form of the 2-view DLT that initialises every triangulation (cv::triangulatePoints,
 * triangulation.cpp:216,290), fixed when the library is built: 2 = rows x*P2-P0, y*P2-P1 per view
 * (4x4 system, later OpenCV releases), 3 = those plus x*P1-y*P0 (6x4, OpenCV 2.4-3.1, the release
 * the reference names). The reference pins no OpenCV version; see DESIGN.md 3. */
int eg3d_dlt_rows(void);

int eg3d_create(const eg3d_scene* scene, int device, eg3d_ctx** out);
void eg3d_destroy(eg3d_ctx* ctx);

/* A second context on the same device that SHARES `parent`'s immutable scene and its currently
 * resident seeds (reference-counted; either may be destroyed first) but has its own HIP stream,
 * events and work buffers. Seeds are independent units on this path
 * (plg_matching_from_refpoints.cpp:83-104 loops over them with no carried state), so a host keeps
 * several batches in flight by driving one context per thread; eg3d_upload_seeds on one context
 * does not affect the others. */
int eg3d_clone(eg3d_ctx* parent, eg3d_ctx** out);

/* which: 0 = 30 px candidate grid, 1 = 4 px expand-all-views grid. Pointers stay
 * valid until eg3d_destroy. cell index = row*ncols + col. (The grids are built and kept on the device; the first call for a
 * cell size copies that grid to the host.) */
int eg3d_get_grid(eg3d_ctx* ctx, int view, int which, uint32_t* ncols, uint32_t* nrows,
                  const uint32_t** cell_off, const uint32_t** ids);

int eg3d_candidates_run(eg3d_ctx* ctx, const eg3d_seeds* seeds, uint32_t seed_begin,
                        uint32_t seed_end, eg3d_candidates* out);
void eg3d_free_candidates(eg3d_candidates* c);

/* Full path on seeds [seed_begin, seed_end). With device_only != 0 the result stays
 * in HBM (out->X etc. are NULL, counts are filled) — used for kernel-only timing. */
int eg3d_match_refpoints(eg3d_ctx* ctx, const eg3d_seeds* seeds, uint32_t seed_begin,
                         uint32_t seed_end, int device_only, eg3d_edgepoints* out,
                         eg3d_stage_times* times /* may be NULL */);
/* Pipelines 1-2 extractor:
 * find_new_3d_points_from_compatible_polylines_expandallviews_parallel
 * (src/edgegraph3d/matching/plg_matching/polyline_matching.cpp:153-208, called per set from
 * pipelines.cpp:98,144) with find_epipolar_correspondences (:45-73). A set of "potentially
 * compatible polylines" is, per view, an ascending list of view-local polyline ids (the
 * reference's vector<set<ulong>>). Every polyline of a set is sampled every 20 px from its start
 * to its end; each sample collects the hits of its epipolar line on the set's polylines of all
 * other views and goes through the same 3-view consensus + expand-all-views as a seed's starting
 * hit. As in the reference's parallel build the PLGMatchesManager is not consulted
 * (is_matched() == false: it is empty when pipelines 1-2 run and never updated inside the loop).
 * Output as eg3d_match_refpoints, in the reference's order (set, start view, polyline id, sample);
 * key = (sample index of the call, start view, 0, index in chain), in the host arrays and in the
 * device view (eg3d_last_device_output after device_only) alike. */
typedef struct eg3d_polyline_sets {
  uint32_t n_sets;
  const uint32_t* row_off; /* [n_sets * n_views + 1] CSR over rows (set * n_views + view) */
  const uint32_t* pl_ids;  /* view-local polyline ids, ascending within a row */
} eg3d_polyline_sets;
int eg3d_match_polyline_sets(eg3d_ctx* ctx, const eg3d_polyline_sets* sets, uint32_t set_begin, uint32_t set_end,
                             int device_only, eg3d_edgepoints* out, eg3d_stage_times* times);

void eg3d_free_edgepoints(eg3d_edgepoints* e);

/* Resident-seed variant: upload once, run many times (bench: inputs in HBM before the
 * timed region). */
int eg3d_upload_seeds(eg3d_ctx* ctx, const eg3d_seeds* seeds);
int eg3d_match_resident(eg3d_ctx* ctx, uint32_t seed_begin, uint32_t seed_end, int device_only,
                        eg3d_edgepoints* out, eg3d_stage_times* times);

/* Internal pipelining of ONE eg3d_match_* call. The reference's parallel entry point runs the seeds of a call on its OpenMP
 * team (plg_matching_from_refpoints.cpp:83-104, `#pragma omp parallel for`; polyline_matching.cpp:153-208 for the sets);
 * here a call's range is cut into `units` contiguous sub-batches (balanced by the sum of track lengths / polylines) that
 * run concurrently on `lanes` internal contexts (own HIP stream, work buffers and host thread, created on first use,
 * shared scene and seeds), so that the candidate / hypothesis stages and the D2H copy of one sub-batch overlap the expand
 * stage of another. The output is the concatenation of the sub-batches in order: byte for byte what a single batch
 * produces. lanes: 1 = no pipelining (the call runs on the context's own stream, one batch of <= 16 384 seeds at a time),
 * 0 = the default, by the kind of call (EG3D_PIPELINE_LANES overrides): 3 for a call that copies its cloud to the host —
 * most of the D2H copy disappears behind the later sub-batches — on a context that has ALREADY completed a host call (lanes
 * cost 50-90 ms, on many-view scenes seconds, when they are created: a one-shot caller's only call runs on the context
 * alone), and 1 for a device-only call, which measured no gain (every expand launch lasts at least as long as its slowest
 * chain: eg3d_stage_times.ms_slowest_chain); units: 0 = chosen
 * from the range (one per lane when each gets >= 128 seeds; EG3D_PIPELINE_UNITS). A clone inherits its parent's setting. Use lanes = 1 on contexts that are themselves driven
 * concurrently (one per host thread): stacking both forms of overlap only multiplies the work buffers. */
int eg3d_set_pipelining(eg3d_ctx* ctx, int lanes, int units);

/* Device-resident view of the edge-points produced by the most recent eg3d_match_* call
 * (pointers into the context's HBM buffers, valid until the next call on this context).
 * A call made with device_only != 0 keeps its WHOLE result in these buffers however many internal
 * sub-batches (<= 16 384 seeds, one expand launch each) it took (global 64-bit observation offsets;
 * obs_off has n_points entries, no sentinel) and `complete` is 1 — this is what the multi-GPU exchange
 * of the edge-point cloud consumes without a host trip; the buffers grow to the size of the call's
 * cloud (12 + 8 + 16 B per point, 20 B per observation). A call that copies to the host
 * (device_only == 0) reuses the buffers per sub-batch: `complete` is then 1 only if it ran as a single one. */
typedef struct eg3d_device_edgepoints {
  uint64_t n_points, n_obs;
  const float* X;
  const uint64_t* obs_off;
  const int32_t* obs_view;
  const uint32_t* obs_pl;
  const uint32_t* obs_seg;
  const float* obs_xy;
  const uint32_t* key;
  int32_t complete;
} eg3d_device_edgepoints;
int eg3d_last_device_output(eg3d_ctx* ctx, eg3d_device_edgepoints* out);

/* Config 5: batched FP32 Gauss-Newton filter. view ids index ctx's cameras.
 * X_out may alias X. legacy_abs != 0 selects the Q9 integer-abs behaviour. */
int eg3d_gn_filter(eg3d_ctx* ctx, const float* X, const uint32_t* obs_off, const int32_t* obs_view,
                   const float* obs_xy, uint64_t n_points, float gn_max_mse, int legacy_abs,
                   float* X_out, uint8_t* inlier, float* ms_kernel /* may be NULL */);

#ifdef __cplusplus
}
#endif
#endif /* EG3D_H_ */
