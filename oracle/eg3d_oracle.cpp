// ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED.
// The reference (abignoli/EdgeGraph3D) has no tests, no golden vectors and cannot be
// compiled here (OpenCV, CGAL, Boost absent: SURVEY.md F3/F4), so this restatement is pinned
// only by hand-derived known-answer tests (tests/test_oracle_kat.py) and by the quirk tests
// Q1-Q15. It is the checker for the HIP path; the product never links or calls it.
//
// Drivers restated here:
//   plg_matching_from_refpoints[_parallel]  src/edgegraph3d/matching/plg_matching/plg_matching_from_refpoints.cpp:83-116
//   gaussNewtonFiltering                    src/edgegraph3d/filtering/gauss_newton.cpp:136-178
//   filter_3d_points_close_2d_array         src/edgegraph3d/filtering/filtering_close_plgps.cpp:99-124
//   compute_inliers (observation filter)    src/edgegraph3d/filtering/outliers_filtering.cpp:37-64
#include "eg3d_oracle.h"

#include <omp.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle_plg.hpp"
#include "oracle_replay.hpp"
#include "oracle_n2.hpp"

using namespace orc;

struct orc_ctx {
  Scene sc;
  std::vector<float> P;
  std::vector<double> F;
  std::vector<uint8_t> Fv;
  // flattened grids for export
  std::vector<std::vector<uint32_t>> g_off[2], g_ids[2];
};

static void flatten_grid(const PolyLine2DMapSearch& g, std::vector<uint32_t>& off, std::vector<uint32_t>& ids) {
  off.assign(g.cells.size() + 1, 0);
  ids.clear();
  for (size_t c = 0; c < g.cells.size(); c++) {
    off[c] = (uint32_t)ids.size();
    for (auto v : g.cells[c]) ids.push_back((uint32_t)v);
  }
  off[g.cells.size()] = (uint32_t)ids.size();
}

extern "C" orc_ctx* orc_create(const eg3d_scene* s) {
  orc_ctx* c = new orc_ctx();
  const int V = s->n_views;
  c->P.assign(s->cam_P, s->cam_P + (size_t)V * 16);
  c->F.assign(s->F, s->F + (size_t)V * V * 9);
  c->Fv.assign(s->F_valid, s->F_valid + (size_t)V * V);
  c->sc.cams.n_views = V;
  c->sc.cams.P = c->P.data();
  c->sc.cams.F = c->F.data();
  c->sc.cams.F_valid = c->Fv.data();
  c->sc.width = s->width;
  c->sc.height = s->height;
  c->sc.dir_mismatch = 0;
  c->sc.plgs.resize(V);
  for (int v = 0; v < V; v++) {
    PLG& g = c->sc.plgs[v];
    uint32_t p0 = s->view_pl_off[v], p1 = s->view_pl_off[v + 1];
    g.polylines.resize(p1 - p0);
    for (uint32_t p = p0; p < p1; p++) {
      polyline& pl = g.polylines[p - p0];
      pl.start = s->pl_start[p];
      pl.end = s->pl_end[p];
      pl.valid = s->pl_valid[p] != 0;
      pl.dir_mismatch_counter = &c->sc.dir_mismatch;
      uint32_t a = s->pl_vtx_off[p], b = s->pl_vtx_off[p + 1];
      if (!pl.valid) b = a;  // an invalidated polyline has no coordinates (polyline_graph_2d.cpp:1047-1058)
      pl.polyline_coords.resize(b - a);
      for (uint32_t i = a; i < b; i++) pl.polyline_coords[i - a] = vec2(s->vtx_xy[2 * i], s->vtx_xy[2 * i + 1]);
    }
  }
  c->sc.grid30.resize(V);
  c->sc.grid4.resize(V);
  for (int w = 0; w < 2; w++) {
    c->g_off[w].resize(V);
    c->g_ids[w].resize(V);
  }
#pragma omp parallel for schedule(dynamic)
  for (int v = 0; v < V; v++) {
    // 30 px: starting_detection_dist*factor (plg_edge_manager.cpp:46,74); 4 px: edge_matcher.cpp:103
    c->sc.grid30[v].build(c->sc.plgs[v], s->width, s->height, 10.0f * 3.0f);
    c->sc.grid4[v].build(c->sc.plgs[v], s->width, s->height, 4.0f);
    flatten_grid(c->sc.grid30[v], c->g_off[0][v], c->g_ids[0][v]);
    flatten_grid(c->sc.grid4[v], c->g_off[1][v], c->g_ids[1][v]);
  }
  return c;
}

extern "C" void orc_destroy(orc_ctx* c) { delete c; }

// DLT form of cv::triangulatePoints: 2 rows per view (4x4, later OpenCV) or 3 (6x4, OpenCV <= 3.1)
extern "C" int orc_set_dlt_rows(int rows) {
  if (rows != 2 && rows != 3) return -1;
  orc::g_dlt_rows = rows;
  return 0;
}
extern "C" int orc_get_dlt_rows(void) { return orc::g_dlt_rows; }
// test hook (tests/test_quirks.py): bit q makes the restatement "fix" quirk Qq (4, 12, 13); 0 = reference behaviour
extern "C" void orc_set_quirk_fixes(unsigned mask) { orc::g_quirk_fix = mask; }
extern "C" void orc_set_conventions(unsigned mask) { orc::g_conv = mask; }

extern "C" int orc_get_grid(orc_ctx* c, int view, int which, uint32_t* ncols, uint32_t* nrows,
                            const uint32_t** cell_off, const uint32_t** ids) {
  if (!c || view < 0 || view >= c->sc.cams.n_views || which < 0 || which > 1) return -1;
  const PolyLine2DMapSearch& g = which == 0 ? c->sc.grid30[view] : c->sc.grid4[view];
  *ncols = g.map_w;
  *nrows = g.map_h;
  *cell_off = c->g_off[which][view].data();
  *ids = c->g_ids[which][view].data();
  return 0;
}

static SeedView seed_view(const eg3d_seeds* s, uint32_t i) {
  SeedView sv;
  sv.views = s->trk_view + s->trk_off[i];
  sv.xy = reinterpret_cast<const vec2*>(s->trk_xy + 2 * (size_t)s->trk_off[i]);
  sv.k = (int)(s->trk_off[i + 1] - s->trk_off[i]);
  return sv;
}

static int pack_edgepoints(orc_ctx* c, std::vector<std::vector<EdgePoint>>& per_seed, std::vector<Stats>& tstats,
                           std::chrono::steady_clock::time_point t0, eg3d_edgepoints* out, orc_stats* stats) {
  auto t1 = std::chrono::steady_clock::now();
  uint64_t np = 0, no = 0;
  for (auto& v : per_seed)
    for (auto& ep : v) {
      np++;
      no += ep.p.obs.size();
    }
  out->n_points = np;
  out->n_obs = no;
  out->X = (float*)malloc(sizeof(float) * 3 * (np ? np : 1));
  out->obs_off = (uint64_t*)malloc(sizeof(uint64_t) * (np + 1));
  out->obs_view = (int32_t*)malloc(sizeof(int32_t) * (no ? no : 1));
  out->obs_pl = (uint32_t*)malloc(sizeof(uint32_t) * (no ? no : 1));
  out->obs_seg = (uint32_t*)malloc(sizeof(uint32_t) * (no ? no : 1));
  out->obs_xy = (float*)malloc(sizeof(float) * 2 * (no ? no : 1));
  out->key = (uint32_t*)malloc(sizeof(uint32_t) * 4 * (np ? np : 1));
  uint64_t pi = 0, oi = 0;
  for (auto& v : per_seed)
    for (auto& ep : v) {
      out->X[3 * pi] = ep.p.X.x;
      out->X[3 * pi + 1] = ep.p.X.y;
      out->X[3 * pi + 2] = ep.p.X.z;
      out->obs_off[pi] = (uint64_t)oi;
      for (int k = 0; k < 4; k++) out->key[4 * pi + k] = ep.key[k];
      for (size_t j = 0; j < ep.p.obs.size(); j++) {
        out->obs_view[oi] = ep.p.views[j];
        out->obs_pl[oi] = (uint32_t)ep.p.obs[j].polyline_id;
        out->obs_seg[oi] = (uint32_t)ep.p.obs[j].plp.segment_index;
        out->obs_xy[2 * oi] = ep.p.obs[j].plp.coords.x;
        out->obs_xy[2 * oi + 1] = ep.p.obs[j].plp.coords.y;
        oi++;
      }
      pi++;
    }
  out->obs_off[np] = (uint64_t)oi;
  Stats tot;
  for (auto& s : tstats) {
    tot.n_tasks += s.n_tasks;
    tot.n_hyp += s.n_hyp;
    tot.n_chains += s.n_chains;
    tot.bytes_algorithmic += s.bytes_algorithmic;
    tot.tri.n_tri += s.tri.n_tri;
    tot.tri.n_add += s.tri.n_add;
    tot.tri.n_degenerate_dlt += s.tri.n_degenerate_dlt;
    tot.tri.n_combos += s.tri.n_combos;
  }
  out->n_tasks = tot.n_tasks;
  out->n_hypotheses = tot.n_hyp;
  out->n_chains = tot.n_chains;
  out->flags = 0;
  if (c->sc.dir_mismatch) out->flags |= EG3D_FLAG_DIR_MISMATCH;
  if (tot.tri.n_degenerate_dlt) out->flags |= EG3D_FLAG_DEGENERATE_DLT;
  if (stats) {
    stats->n_tasks = tot.n_tasks;
    stats->n_hyp = tot.n_hyp;
    stats->n_chains = tot.n_chains;
    stats->n_tri = tot.tri.n_tri;
    stats->n_add = tot.tri.n_add;
    stats->n_degenerate_dlt = tot.tri.n_degenerate_dlt;
    stats->n_combos = tot.tri.n_combos;
    stats->bytes_algorithmic = tot.bytes_algorithmic;
    stats->dir_mismatch = c->sc.dir_mismatch;
    uint32_t gd = 0;
    for (auto& g : c->sc.grid30) gd += g.dropped_out_of_range;
    for (auto& g : c->sc.grid4) gd += g.dropped_out_of_range;
    stats->grid_dropped = gd;
    stats->seconds = std::chrono::duration<double>(t1 - t0).count();
  }
  return 0;
}

extern "C" int orc_match_refpoints(orc_ctx* c, const eg3d_seeds* seeds, uint32_t b, uint32_t e, int nthreads,
                                   eg3d_edgepoints* out, orc_stats* stats) {
  if (!c || !seeds || e > seeds->n_seeds || b > e) return -1;
  memset(out, 0, sizeof(*out));
  const uint32_t n = e - b;
  std::vector<std::vector<EdgePoint>> per_seed(n);
  if (nthreads < 1) nthreads = 1;
  std::vector<Stats> tstats(nthreads);
  c->sc.dir_mismatch = 0;
  auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (uint32_t i = 0; i < n; i++) {
    int t = omp_get_thread_num();
    SeedView sv = seed_view(seeds, b + i);
    plg_matching_from_refpoint(c->sc, sv, b + i, per_seed[i], &tstats[t]);
  }
  return pack_edgepoints(c, per_seed, tstats, t0, out, stats);
}

extern "C" int orc_match_polyline_sets(orc_ctx* c, uint32_t n_sets, const uint32_t* row_off, const uint32_t* pl_ids,
                                       uint32_t b, uint32_t e, int nthreads, eg3d_edgepoints* out, orc_stats* stats) {
  if (!c || !row_off || !out || e > n_sets || b > e) return -1;
  memset(out, 0, sizeof(*out));
  const int V = (int)c->sc.plgs.size();
  const uint32_t n = e - b;
  std::vector<std::vector<std::vector<ulong_t>>> sets(n, std::vector<std::vector<ulong_t>>(V));
  for (uint32_t i = 0; i < n; i++)
    for (int v = 0; v < V; v++) {
      const uint32_t row = (b + i) * (uint32_t)V + (uint32_t)v;
      for (uint32_t k = row_off[row]; k < row_off[row + 1]; k++) sets[i][v].push_back(pl_ids[k]);
    }
  std::vector<uint32_t> base(n + 1, 0);
  for (uint32_t i = 0; i < n; i++) base[i + 1] = base[i] + count_set_samples(c->sc, sets[i]);
  std::vector<std::vector<EdgePoint>> per_set(n);
  if (nthreads < 1) nthreads = 1;
  std::vector<Stats> tstats(nthreads);
  c->sc.dir_mismatch = 0;
  auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (uint32_t i = 0; i < n; i++) {
    int t = omp_get_thread_num();
    match_polyline_set(c->sc, sets[i], base[i], per_set[i], &tstats[t]);
  }
  return pack_edgepoints(c, per_set, tstats, t0, out, stats);
}

extern "C" void orc_free_edgepoints(eg3d_edgepoints* e) {
  if (!e) return;
  free(e->X);
  free(e->obs_off);
  free(e->obs_view);
  free(e->obs_pl);
  free(e->obs_seg);
  free(e->obs_xy);
  free(e->key);
  memset(e, 0, sizeof(*e));
}

template <typename T>
static T* dup_vec(const std::vector<T>& v) {
  T* p = (T*)malloc(sizeof(T) * (v.size() ? v.size() : 1));
  if (!v.empty()) memcpy(p, v.data(), sizeof(T) * v.size());
  return p;
}

extern "C" int orc_candidates(orc_ctx* c, const eg3d_seeds* seeds, uint32_t b, uint32_t e, eg3d_candidates* out) {
  if (!c || !seeds || e > seeds->n_seeds || b > e) return -1;
  memset(out, 0, sizeof(*out));
  std::vector<uint32_t> cand_off(1, 0), cand_pl, start_off(1, 0), start_pl, start_seg, task_sv, task_hit,
      task_list_off(1, 0), list_off(1, 0), hit_pl, hit_seg;
  std::vector<float> start_xy, hit_xy;
  uint32_t sv_base = 0;
  for (uint32_t s = b; s < e; s++) {
    SeedView sv = seed_view(seeds, s);
    StageA sa = detect_nearby_intersections_and_correspondences_plgp(c->sc, sv, nullptr);
    for (int a = 0; a < sv.k; a++) {
      for (auto id : sa.cand[a]) cand_pl.push_back((uint32_t)id);
      cand_off.push_back((uint32_t)cand_pl.size());
      for (auto& h : sa.start_hits[a]) {
        start_pl.push_back((uint32_t)h.polyline_id);
        start_seg.push_back((uint32_t)h.plp.segment_index);
        start_xy.push_back(h.plp.coords.x);
        start_xy.push_back(h.plp.coords.y);
      }
      start_off.push_back((uint32_t)start_pl.size());
    }
    for (int a = 0; a < sv.k; a++)
      for (size_t h = 0; h < sa.start_hits[a].size(); h++) {
        task_sv.push_back(sv_base + a);
        task_hit.push_back((uint32_t)h);
        for (int i = 0; i < sv.k; i++) {
          for (auto& p : sa.corr[a][h][i]) {
            hit_pl.push_back((uint32_t)p.polyline_id);
            hit_seg.push_back((uint32_t)p.plp.segment_index);
            hit_xy.push_back(p.plp.coords.x);
            hit_xy.push_back(p.plp.coords.y);
          }
          list_off.push_back((uint32_t)hit_pl.size());
        }
        task_list_off.push_back((uint32_t)list_off.size() - 1);
      }
    sv_base += sv.k;
  }
  out->n_sv = sv_base;
  out->cand_off = dup_vec(cand_off);
  out->cand_pl = dup_vec(cand_pl);
  out->start_off = dup_vec(start_off);
  out->start_pl = dup_vec(start_pl);
  out->start_seg = dup_vec(start_seg);
  out->start_xy = dup_vec(start_xy);
  out->n_tasks = (uint32_t)task_sv.size();
  out->task_sv = dup_vec(task_sv);
  out->task_hit = dup_vec(task_hit);
  out->task_list_off = dup_vec(task_list_off);
  out->list_off = dup_vec(list_off);
  out->hit_pl = dup_vec(hit_pl);
  out->hit_seg = dup_vec(hit_seg);
  out->hit_xy = dup_vec(hit_xy);
  return 0;
}

extern "C" void orc_free_candidates(eg3d_candidates* c) {
  if (!c) return;
  free(c->cand_off);
  free(c->cand_pl);
  free(c->start_off);
  free(c->start_pl);
  free(c->start_seg);
  free(c->start_xy);
  free(c->task_sv);
  free(c->task_hit);
  free(c->task_list_off);
  free(c->list_off);
  free(c->hit_pl);
  free(c->hit_seg);
  free(c->hit_xy);
  memset(c, 0, sizeof(*c));
}

// gaussNewtonFiltering, gauss_newton.cpp:136-178
extern "C" int orc_gn_filter(orc_ctx* c, const float* X, const uint32_t* obs_off, const int32_t* obs_view,
                             const float* obs_xy, uint64_t n_points, float gn_max_mse, int legacy_abs, int nthreads,
                             float* X_out, uint8_t* inlier) {
  if (!c) return -1;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (uint64_t i = 0; i < n_points; i++) {
    uint32_t a = obs_off[i], b = obs_off[i + 1];
    std::vector<GNObs> obs(b - a);
    for (uint32_t j = a; j < b; j++) {
      obs[j - a].P = c->P.data() + (size_t)obs_view[j] * 16;
      obs[j - a].x = obs_xy[2 * j];
      obs[j - a].y = obs_xy[2 * j + 1];
    }
    float init[3] = {X[3 * i], X[3 * i + 1], X[3 * i + 2]}, opt[3];
    if (GaussNewton_f32(obs, init, opt, gn_max_mse, legacy_abs != 0) != -1) {
      X_out[3 * i] = opt[0];
      X_out[3 * i + 1] = opt[1];
      X_out[3 * i + 2] = opt[2];
      inlier[i] = 1;
    } else {
      X_out[3 * i] = init[0];
      X_out[3 * i + 1] = init[1];
      X_out[3 * i + 2] = init[2];
      inlier[i] = 0;
    }
  }
  return 0;
}

// filter_3d_points_close_2d_array, filtering_close_plgps.cpp:75-124 (CELLSIZE 3)
extern "C" int orc_filter_close_2d(orc_ctx* c, const eg3d_edgepoints* pts, uint8_t* keep) {
  if (!c) return -1;
  const int CELL = 3;
  const int w = int(std::ceil((float)c->sc.width / CELL)), h = int(std::ceil((float)c->sc.height / CELL));
  const int V = c->sc.cams.n_views;
  std::vector<std::vector<uint8_t>> bmaps(V, std::vector<uint8_t>((size_t)w * h, 0));
  // An observation outside the maps (coordinates at/past the border, negative beyond one cell, NaN,
  // view id out of range) indexes out of bounds in the reference — undefined behaviour. The rule
  // adopted by oracle and product alike: it never makes a point new and marks nothing.
  auto inside = [&](uint32_t j, int& cx, int& cy) {
    const int v = pts->obs_view[j];
    const float fx = pts->obs_xy[2 * j] / CELL, fy = pts->obs_xy[2 * j + 1] / CELL;
    if (v < 0 || v >= V || !(fx > -1.0f) || !(fy > -1.0f) || !(fx < (float)w) || !(fy < (float)h)) return false;
    cx = int(fx);
    cy = int(fy);
    return cx >= 0 && cy >= 0 && cx < w && cy < h;
  };
  for (uint64_t i = 0; i < pts->n_points; i++) {
    bool is_new = false;
    for (uint32_t j = pts->obs_off[i]; j < pts->obs_off[i + 1]; j++) {
      int cx, cy;
      if (!inside(j, cx, cy)) continue;
      if (!bmaps[pts->obs_view[j]][(size_t)cy * w + cx]) {
        is_new = true;
        break;
      }
    }
    keep[i] = is_new ? 1 : 0;
    if (is_new)
      for (uint32_t j = pts->obs_off[i]; j < pts->obs_off[i + 1]; j++) {
        int cx, cy;
        if (inside(j, cx, cy)) bmaps[pts->obs_view[j]][(size_t)cy * w + cx] = 1;
      }
  }
  return 0;
}

// Row a17: replay of plgmm.add_matched_3dpolyline over the chains of an edge-point cloud
// (plg_matching_from_refpoints.cpp:74-77). Output in the eg3d_graph3d layout of include/eg3d_host.h.
extern "C" int orc_replay_matches(orc_ctx* c, const eg3d_edgepoints* pts, eg3d_graph3d* out) {
  if (!c || !pts || !out) return -1;
  memset(out, 0, sizeof(*out));
  MatchesManager mm(c->sc);
  std::vector<MatchesManager::P3> chain;
  auto flush = [&]() {
    if (!chain.empty()) mm.add_matched_3dpolyline(chain);
    chain.clear();
  };
  for (uint64_t i = 0; i < pts->n_points; i++) {
    const uint32_t* k = pts->key + 4 * i;
    if (i > 0) {
      const uint32_t* kp = pts->key + 4 * (i - 1);
      if (!(kp[0] == k[0] && kp[1] == k[1] && kp[2] == k[2] && k[3] == kp[3] + 1)) flush();
    }
    MatchesManager::P3 p;
    p.X = vec3r{pts->X[3 * i], pts->X[3 * i + 1], pts->X[3 * i + 2]};
    p.index = i;
    for (uint32_t o = pts->obs_off[i]; o < pts->obs_off[i + 1]; o++) {
      p.obs.push_back(plg_point(pts->obs_pl[o], pts->obs_seg[o], vec2(pts->obs_xy[2 * o], pts->obs_xy[2 * o + 1])));
      p.views.push_back(pts->obs_view[o]);
    }
    chain.push_back(std::move(p));
  }
  flush();
  const PLG3D& g = mm.plg3d;
  const uint64_t NN = g.nodes_coords.size();
  out->n_nodes = NN;
  out->n_real_nodes = g.real_nodes_amount;
  out->n_polylines = g.polylines.size();
  out->node_X = (float*)malloc(sizeof(float) * 3 * (NN + 1));
  out->node_point = (uint64_t*)malloc(sizeof(uint64_t) * (NN + 1));
  out->conn_off = (uint64_t*)malloc(sizeof(uint64_t) * (NN + 1));
  uint64_t nconn = 0;
  for (uint64_t n = 0; n < NN; n++) nconn += g.connections[n].size();
  out->conn_pl = (uint32_t*)malloc(sizeof(uint32_t) * (nconn + 1));
  uint64_t w = 0;
  for (uint64_t n = 0; n < NN; n++) {
    out->node_X[3 * n] = g.nodes_coords[n].x;
    out->node_X[3 * n + 1] = g.nodes_coords[n].y;
    out->node_X[3 * n + 2] = g.nodes_coords[n].z;
    out->node_point[n] = g.node_point[n];
    out->conn_off[n] = w;
    for (auto p : g.connections[n]) out->conn_pl[w++] = (uint32_t)p;
  }
  out->conn_off[NN] = w;
  out->pl_start = (uint32_t*)malloc(sizeof(uint32_t) * (g.polylines.size() + 1));
  out->pl_end = (uint32_t*)malloc(sizeof(uint32_t) * (g.polylines.size() + 1));
  for (size_t p = 0; p < g.polylines.size(); p++) {
    out->pl_start[p] = (uint32_t)g.polylines[p].start;
    out->pl_end[p] = (uint32_t)g.polylines[p].end;
  }
  uint64_t NP = 0, NI = 0;
  for (auto& v : mm.matched) {
    NP += v.size();
    for (auto& S : v) NI += S.size();
  }
  out->n_scene_polylines = NP;
  out->iv_off = (uint64_t*)malloc(sizeof(uint64_t) * (NP + 1));
  out->iv_start_seg = (uint32_t*)malloc(sizeof(uint32_t) * (NI + 1));
  out->iv_end_seg = (uint32_t*)malloc(sizeof(uint32_t) * (NI + 1));
  out->iv_start_xy = (float*)malloc(sizeof(float) * 2 * (NI + 1));
  out->iv_end_xy = (float*)malloc(sizeof(float) * 2 * (NI + 1));
  uint64_t gp = 0, k = 0;
  for (auto& v : mm.matched)
    for (auto& S : v) {
      out->iv_off[gp++] = k;
      for (auto& iv : S) {
        out->iv_start_seg[k] = (uint32_t)iv.start.segment_index;
        out->iv_end_seg[k] = (uint32_t)iv.end.segment_index;
        out->iv_start_xy[2 * k] = iv.start.coords.x;
        out->iv_start_xy[2 * k + 1] = iv.start.coords.y;
        out->iv_end_xy[2 * k] = iv.end.coords.x;
        out->iv_end_xy[2 * k + 1] = iv.end.coords.y;
        k++;
      }
    }
  out->iv_off[NP] = k;
  return 0;
}
extern "C" void orc_free_graph3d(eg3d_graph3d* g) {
  if (!g) return;
  free(g->node_X);
  free(g->node_point);
  free(g->pl_start);
  free(g->pl_end);
  free(g->conn_off);
  free(g->conn_pl);
  free(g->iv_off);
  free(g->iv_start_seg);
  free(g->iv_end_seg);
  free(g->iv_start_xy);
  free(g->iv_end_xy);
  memset(g, 0, sizeof(*g));
}

// one binary edge image -> optimised polyline graph (convertEdgeImagePolyLineGraph_optimized,
// convert_edge_images_pixel_to_segment.cpp:879-883). Output in the eg3d_plg_view layout; free with orc_free_plg_view.
extern "C" int orc_plg_from_mask(const uint8_t* mask_in, int width, int height, eg3d_plg_view* out) {
  if (!mask_in || !out || width <= 0 || height <= 0) return -1;
  std::vector<uint8_t> mask(mask_in, mask_in + (size_t)width * height);
  for (auto& m : mask) m = m ? 1 : 0;
  orc::n2::PLG2 g;
  orc::n2::edge_image_to_plg(mask.data(), height, width, g);
  memset(out, 0, sizeof(*out));
  const size_t NP = g.polylines.size(), NN = g.nodes_coords.size();
  out->n_polylines = (uint32_t)NP;
  out->n_nodes = (uint32_t)NN;
  out->pl_vtx_off = (uint32_t*)malloc(sizeof(uint32_t) * (NP + 1));
  out->pl_start = (uint32_t*)malloc(sizeof(uint32_t) * (NP + 1));
  out->pl_end = (uint32_t*)malloc(sizeof(uint32_t) * (NP + 1));
  out->pl_valid = (uint8_t*)malloc(NP + 1);
  size_t nv = 0;
  for (size_t p = 0; p < NP; p++) nv += g.polylines[p].coords.size();
  out->vtx_xy = (float*)malloc(sizeof(float) * 2 * (nv + 1));
  out->node_xy = (float*)malloc(sizeof(float) * 2 * (NN + 1));
  size_t w = 0;
  for (size_t p = 0; p < NP; p++) {
    out->pl_vtx_off[p] = (uint32_t)w;
    out->pl_start[p] = (uint32_t)g.polylines[p].start;
    out->pl_end[p] = (uint32_t)g.polylines[p].end;
    out->pl_valid[p] = g.is_valid_polyline(p) ? 1 : 0;
    for (auto& c : g.polylines[p].coords) {
      out->vtx_xy[2 * w] = c.x;
      out->vtx_xy[2 * w + 1] = c.y;
      w++;
    }
  }
  out->pl_vtx_off[NP] = (uint32_t)w;
  for (size_t n = 0; n < NN; n++) {
    out->node_xy[2 * n] = g.nodes_coords[n].x;
    out->node_xy[2 * n + 1] = g.nodes_coords[n].y;
  }
  return 0;
}
extern "C" void orc_free_plg_view(eg3d_plg_view* v) {
  if (!v) return;
  free(v->pl_vtx_off);
  free(v->vtx_xy);
  free(v->pl_start);
  free(v->pl_end);
  free(v->pl_valid);
  free(v->node_xy);
  memset(v, 0, sizeof(*v));
}

// compute_ray_stats + compute_inliers tail, outliers_filtering.cpp:14-35,37-64
extern "C" int orc_observation_filter(int n_cameras, const uint32_t* obs_off, uint64_t n_points,
                                      uint64_t first_edgepoint, int forced_min_filter, uint8_t* inlier) {
  std::vector<int> dist(n_cameras, 0);
  int count = 0;
  for (uint64_t i = 0; i < n_points; i++)
    if (inlier[i]) {
      count++;
      int k = (int)(obs_off[i + 1] - obs_off[i]);
      if (k >= 1 && k <= n_cameras) dist[k - 1]++;
    }
  int m_amount = 0, median = 0;
  for (median = 0; median < n_cameras; median++) {
    m_amount += dist[median];
    if (m_amount >= count / 2) break;
  }
  int intended = (3 >= median / 2 - 1) ? 3 : (median / 2 - 1);
  if (forced_min_filter > -1) intended = forced_min_filter;
  for (uint64_t i = first_edgepoint; i < n_points; i++)
    inlier[i] = inlier[i] && ((int)(obs_off[i + 1] - obs_off[i]) > intended);
  return intended;
}

// ------------------------------------------------------------- primitive probes ----
extern "C" float orc_squared_2d_distance(float ax, float ay, float bx, float by) {
  return squared_2d_distance(vec2(ax, ay), vec2(bx, by));
}
extern "C" float orc_minimum_distancesq(float px, float py, float vx, float vy, float wx, float wy, float* proj) {
  vec2 pr;
  float d = minimum_distancesq(vec2(px, py), vec2(vx, vy), vec2(wx, wy), pr);
  proj[0] = pr.x;
  proj[1] = pr.y;
  return d;
}
extern "C" int orc_intersect_segment_line(float x1, float y1, float x2, float y2, const float* line, float* inter,
                                          int* parallel, int* overlapped) {
  bool p, o, f;
  vec2 i;
  intersect_segment_line(x1, y1, x2, y2, line, p, o, f, i);
  inter[0] = i.x;
  inter[1] = i.y;
  *parallel = p;
  *overlapped = o;
  return f;
}
extern "C" int orc_intersect_segment_line_nqp(float x1, float y1, float x2, float y2, const float* line, float* inter,
                                              int* quasiparallel, float* distance) {
  bool p, o, f, q, v;
  vec2 i;
  float d = -1;
  intersect_segment_line_no_quasiparallel(x1, y1, x2, y2, line, QP_COS, QP_DIST, p, o, f, q, v, d, i);
  inter[0] = i.x;
  inter[1] = i.y;
  *quasiparallel = q;
  *distance = d;
  return f;
}
extern "C" int orc_cell_from_coords(float cell, float x, float y, int* col, int* row, int* b_row, int* b_col) {
  bool br, bc;
  auto c = get_2dmap_cell_from_coords(cell, vec2(x, y), br, bc);
  *col = (int)c.first;
  *row = (int)c.second;
  *b_row = br;
  *b_col = bc;
  return 0;
}
extern "C" int orc_epiline(const double* F9, float x, float y, float* line) {
  Cameras cm;
  uint8_t one = 1;
  cm.n_views = 1;
  cm.F = F9;
  cm.F_valid = &one;
  cm.P = nullptr;
  return computeCorrespondEpilineSinglePoint(cm, 0, 0, vec2(x, y), line);
}
// batched forms of the probes the glm pin test drives with ~1 M cases (tests/test_glm_pin.py)
extern "C" void orc_batch_project(uint64_t n, const float* P16, const float* X, float* xy) {
  for (uint64_t i = 0; i < n; i++) {
    vec2 r = compute_projection(P16 + 16 * i, vec3(X[3 * i], X[3 * i + 1], X[3 * i + 2]));
    xy[2 * i] = r.x;
    xy[2 * i + 1] = r.y;
  }
}
extern "C" void orc_batch_mindist(uint64_t n, const float* pvw /*[n][6]*/, float* out /*[n][3]: d2, proj*/) {
  for (uint64_t i = 0; i < n; i++) {
    const float* q = pvw + 6 * i;
    vec2 pr;
    out[3 * i] = minimum_distancesq(vec2(q[0], q[1]), vec2(q[2], q[3]), vec2(q[4], q[5]), pr);
    out[3 * i + 1] = pr.x;
    out[3 * i + 2] = pr.y;
  }
}
extern "C" void orc_batch_anglecos(uint64_t n, const float* seg_line /*[n][7]: x1 y1 x2 y2 a b c*/, float* out) {
  for (uint64_t i = 0; i < n; i++) {
    const float* q = seg_line + 7 * i;
    out[i] = compute_anglecos(q[0], q[1], q[2], q[3], q + 4);
  }
}
extern "C" void orc_project(const float* P16, const float* X, float* xy) {
  vec2 r = compute_projection(P16, vec3(X[0], X[1], X[2]));
  xy[0] = r.x;
  xy[1] = r.y;
}
extern "C" int orc_triangulate(const float* P, const int* view_ids, const float* xy, int n, float* X,
                               int* degenerate) {
  Cameras cm;
  cm.P = P;
  cm.F = nullptr;
  cm.F_valid = nullptr;
  cm.n_views = 0;
  std::vector<vec2> coords(n);
  std::vector<int> ids(view_ids, view_ids + n);
  for (int i = 0; i < n; i++) coords[i] = vec2(xy[2 * i], xy[2 * i + 1]);
  vec3 out;
  bool valid;
  TriStats st;
  em_estimate3Dpositions(cm, coords, ids, out, valid, &st);
  if (degenerate) *degenerate = (int)st.n_degenerate_dlt;
  if (valid) {
    X[0] = out.x;
    X[1] = out.y;
    X[2] = out.z;
  }
  return valid;
}
extern "C" int orc_gn_add(const float* P, const int* view_ids, const float* xy, int n, const float* X0, float* X) {
  Cameras cm;
  cm.P = P;
  cm.F = nullptr;
  cm.F_valid = nullptr;
  cm.n_views = 0;
  std::vector<vec2> coords(n - 1);
  std::vector<int> ids(view_ids, view_ids + n - 1);
  for (int i = 0; i < n - 1; i++) coords[i] = vec2(xy[2 * i], xy[2 * i + 1]);
  vec3 out;
  bool valid;
  em_add_new_observation_to_3Dpositions(cm, vec3(X0[0], X0[1], X0[2]), coords, ids,
                                        vec2(xy[2 * (n - 1)], xy[2 * (n - 1) + 1]), view_ids[n - 1], out, valid,
                                        nullptr);
  if (valid) {
    X[0] = out.x;
    X[1] = out.y;
    X[2] = out.z;
  }
  return valid;
}
extern "C" void orc_dlt(const float* P1, const float* xy1, const float* P2, const float* xy2, double* X0) {
  dlt2_init(P1, vec2(xy1[0], xy1[1]), P2, vec2(xy2[0], xy2[1]), X0);
}
static polyline make_pl(const float* vtx, int n, uint32_t start, uint32_t end) {
  polyline pl;
  pl.start = start;
  pl.end = end;
  pl.valid = true;
  pl.polyline_coords.resize(n);
  for (int i = 0; i < n; i++) pl.polyline_coords[i] = vec2(vtx[2 * i], vtx[2 * i + 1]);
  return pl;
}
extern "C" int orc_next_by_distance(const float* vtx, int n, uint32_t start, uint32_t end, uint32_t seg, float x,
                                    float y, uint32_t direction, float distance, uint32_t* out_seg, float* out_xy) {
  polyline pl = make_pl(vtx, n, start, end);
  bool reached;
  pl_point r = pl.next_pl_point_by_distance(pl_point(seg, vec2(x, y)), direction, distance, reached);
  *out_seg = (uint32_t)r.segment_index;
  out_xy[0] = r.coords.x;
  out_xy[1] = r.coords.y;
  return reached;
}
extern "C" int orc_next_by_line(const float* vtx, int n, uint32_t start, uint32_t end, uint32_t seg, float x, float y,
                                uint32_t direction, const float* line, int bounded, float mind, float maxd,
                                uint32_t* out_seg, float* out_xy, int* flags) {
  polyline pl = make_pl(vtx, n, start, end);
  pl_point next, nbq;
  bool fqp, reached, bdv, found;
  pl.next_pl_point_by_line_intersection_impl(pl_point(seg, vec2(x, y)), direction, line, bounded != 0, mind, maxd,
                                             next, fqp, nbq, reached, bdv, found);
  *out_seg = (uint32_t)next.segment_index;
  out_xy[0] = next.coords.x;
  out_xy[1] = next.coords.y;
  *flags = (fqp ? 1 : 0) | (reached ? 2 : 0) | (bdv ? 4 : 0);
  return found;
}

// ---------------------------------------------------------------------------- N4 ----
// get_point_sets_on_images (edge_graph_3d_utilities.cpp:369-380), find_points_on_both_images (:345-352),
// get_2d_coordinates_of_point_on_image (:382-393), findFundamentalMatrixFromPoints (geometric_utilities.cpp:754-779)
#include <set>
extern "C" int orc_pair_correspondences(int n_views, uint64_t n_points, const uint32_t* trk_off, const int32_t* trk_view,
                                        const float* trk_xy, int i, int j, uint32_t* ids, float* xy_i, float* xy_j,
                                        uint32_t* n_common) {
  // pointsVisibleFromCamN_ as OpenMvgParser fills it: one entry per observation, then copied into a set
  std::vector<std::set<int>> points_on_images((size_t)n_views);
  for (uint64_t p = 0; p < n_points; p++)
    for (uint32_t k = trk_off[p]; k < trk_off[p + 1]; k++)
      if (trk_view[k] >= 0 && trk_view[k] < n_views) points_on_images[(size_t)trk_view[k]].insert((int)p);
  std::set<int> both;
  for (int id : points_on_images[(size_t)i])
    if (points_on_images[(size_t)j].count(id)) both.insert(id);
  if (n_common) *n_common = (uint32_t)both.size();
  if (both.size() < 10) return 0;
  int n = 0;
  for (int id : both) {
    float a[2] = {0, 0}, b[2] = {0, 0};
    for (uint32_t k = trk_off[id]; k < trk_off[id + 1]; k++) {
      if (trk_view[k] == i) a[0] = trk_xy[2 * k], a[1] = trk_xy[2 * k + 1];
      if (trk_view[k] == j) b[0] = trk_xy[2 * k], b[1] = trk_xy[2 * k + 1];
    }
    ids[n] = (uint32_t)id;
    xy_i[2 * n] = a[0], xy_i[2 * n + 1] = a[1];
    xy_j[2 * n] = b[0], xy_j[2 * n + 1] = b[1];
    n++;
  }
  return n;
}
