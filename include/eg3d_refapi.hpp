// eg3d_refapi.hpp — header-only C++ shim that offers the reference's call surface for the hot
// path on top of the C ABI (include/eg3d.h), so reference-side host code can switch with a
// type alias and without touching its callers. It mirrors, name for name:
//
//   plg_matching_from_refpoints_parallel(sfmd, em, cm, plgmm)
//        include/edgegraph3d/matching/plg_matching/plg_matching_from_refpoints.hpp:53,55
//   PLGEdgeManager(imgs, sfmd, F, plgs, 10, 3)  +  detect_nearby_intersections_and_correspondences_plgp(int)
//        include/edgegraph3d/edge_managers/plg_edge_manager.hpp:74,80
//   find_new_3d_points_from_compatible_polylines_expandallviews_parallel(sfmd, ..., potentially_compatible_polylines, ...)
//        include/edgegraph3d/matching/plg_matching/polyline_matching.hpp:55-56
//   gaussNewtonFiltering(SfMData&, std::vector<bool>&, float)
//        include/edgegraph3d/filtering/gauss_newton.hpp:20
//
// Types: the structs below have the fields of the reference types the path reads
// (SfMData.h:16-30, types_reconstructor.hpp:68-82, polyline_graph_2d.hpp:85-119,222-294) with
// plain float arrays in place of glm/cv::Mat. Error behaviour follows the reference: no
// exceptions on the path, an empty result on failure (the C ABI status is available through
// last_status()).
#pragma once
#include <array>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "eg3d.h"
#include "eg3d_host.h" /* eg3d_host_replay_matches (libeg3d_host.so) */

namespace eg3d_ref {

struct vec2 {
  float x, y;
};
struct vec3 {
  float x, y, z;
};
struct CameraType {
  float cameraMatrix[4][4];  // cameraMatrix[r][c], last row zero (OpenMvgParser.cpp:107-125)
};
struct SfMData {
  int numPoints_ = 0, numCameras_ = 0;
  std::vector<vec3> points_;
  std::vector<CameraType> camerasList_;
  std::vector<std::vector<int>> camViewingPointN_;
  std::vector<std::vector<vec2>> point2DoncamViewingPoint_;
  int imageWidth_ = 0, imageHeight_ = 0;
};
struct PolyLineGraph2D {
  struct polyline {
    unsigned long start = 0, end = 0;
    std::vector<vec2> polyline_coords;
  };
  struct pl_point {
    unsigned long segment_index;
    vec2 coords;
  };
  struct plg_point {
    unsigned long polyline_id;
    pl_point plp;
  };
  std::vector<polyline> polylines;
  std::vector<vec2> nodes_coords;
  // PolyLineGraph2D::is_valid_polyline (polyline_graph_2d.cpp:1141-1147)
  bool is_valid_polyline(size_t id) const {
    const polyline& p = polylines[id];
    auto valid_node = [&](unsigned long n) {
      return n < nodes_coords.size() && nodes_coords[n].x != -1 && nodes_coords[n].y != -1;
    };
    if (!valid_node(p.start) || !valid_node(p.end) || p.polyline_coords.size() <= 1) return false;
    const vec2 &a = nodes_coords[p.start], &b = nodes_coords[p.end];
    const vec2 &f = p.polyline_coords.front(), &l = p.polyline_coords.back();
    return a.x == f.x && a.y == f.y && b.x == l.x && b.y == l.y;
  }
};
using FundamentalMatrices = std::vector<std::vector<std::array<double, 9>>>;  // F[i][j]; all-zero => invalid (1x1 Mat)

using new_3dpoint_plgp_matches = std::tuple<vec3, std::vector<PolyLineGraph2D::plg_point>, std::vector<int>>;

// PLGMatchesManager of the path (plg_matches_manager.hpp:99-141): what add_matched_3dpolyline leaves
// behind — the 3-D polyline graph (get_plg3d()) and the matched 2-D intervals — rebuilt by
// eg3d_host_replay_matches (include/eg3d_host.h, row a17) from the ordered output of a run.
class PLGMatchesManager {
 public:
  PLGMatchesManager() { std::memset(&g_, 0, sizeof(g_)); }
  ~PLGMatchesManager() { eg3d_host_free_graph3d(&g_); }
  PLGMatchesManager(const PLGMatchesManager&) = delete;
  PLGMatchesManager& operator=(const PLGMatchesManager&) = delete;
  const eg3d_graph3d& get_plg3d() const { return g_; }
  // matched_polyline_intervals[plg_id][polyline_id] as [begin, end) into the iv_* arrays of get_plg3d()
  std::pair<uint64_t, uint64_t> matched_intervals(const eg3d_scene& sc, int plg_id, unsigned long polyline_id) const {
    const uint64_t gpl = sc.view_pl_off[plg_id] + polyline_id;
    return {g_.iv_off[gpl], g_.iv_off[gpl + 1]};
  }
  // concatenates the batches of a run (seed order) and replays them
  int replay(const eg3d_scene& sc, const std::vector<eg3d_edgepoints>& parts) {
    eg3d_host_free_graph3d(&g_);
    std::vector<float> X, xy;
    std::vector<uint32_t> pl, seg, key;
    std::vector<uint64_t> off(1, 0);
    std::vector<int32_t> view;
    for (const eg3d_edgepoints& e : parts) {
      const uint64_t base = (uint64_t)view.size();
      X.insert(X.end(), e.X, e.X + 3 * e.n_points);
      key.insert(key.end(), e.key, e.key + 4 * e.n_points);
      for (uint64_t i = 0; i < e.n_points; i++) off.push_back(base + e.obs_off[i + 1]);
      view.insert(view.end(), e.obs_view, e.obs_view + e.n_obs);
      pl.insert(pl.end(), e.obs_pl, e.obs_pl + e.n_obs);
      seg.insert(seg.end(), e.obs_seg, e.obs_seg + e.n_obs);
      xy.insert(xy.end(), e.obs_xy, e.obs_xy + 2 * e.n_obs);
    }
    eg3d_edgepoints all;
    std::memset(&all, 0, sizeof(all));
    all.n_points = off.size() - 1;
    all.n_obs = view.size();
    all.X = X.data();
    all.obs_off = off.data();
    all.obs_view = view.data();
    all.obs_pl = pl.data();
    all.obs_seg = seg.data();
    all.obs_xy = xy.data();
    all.key = key.data();
    return eg3d_host_replay_matches(&sc, &all, &g_);
  }

 private:
  eg3d_graph3d g_;
};

// Owns the flattened scene and the GPU context: the PLGEdgeManager + PLGPCM3ViewsPLGFollowing
// pair of the reference collapsed into one object (edge_matcher.cpp:101-115).
class PLGEdgeManager {
 public:
  PLGEdgeManager(const SfMData& sfmd, const FundamentalMatrices& F, const std::vector<PolyLineGraph2D>& plgs,
                 int device = 0)
      : sfmd_(sfmd) {
    const int V = sfmd.numCameras_;
    for (int v = 0; v < V; v++)
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) P_.push_back(sfmd.camerasList_[v].cameraMatrix[r][c]);
    F_.assign((size_t)V * V * 9, 0.0);
    Fv_.assign((size_t)V * V, 0);
    for (int i = 0; i < V; i++)
      for (int j = 0; j < V; j++) {
        bool any = false;
        for (int k = 0; k < 9; k++) {
          F_[((size_t)i * V + j) * 9 + k] = F[i][j][k];
          any = any || F[i][j][k] != 0.0;
        }
        Fv_[(size_t)i * V + j] = (i != j && any) ? 1 : 0;
      }
    vpo_.push_back(0);
    pvo_.push_back(0);
    for (int v = 0; v < V; v++) {
      for (size_t p = 0; p < plgs[v].polylines.size(); p++) {
        const auto& pl = plgs[v].polylines[p];
        const bool ok = plgs[v].is_valid_polyline(p);
        pls_.push_back((uint32_t)pl.start);
        ple_.push_back((uint32_t)pl.end);
        plv_.push_back(ok ? 1 : 0);
        if (ok)
          for (const auto& c : pl.polyline_coords) {
            vtx_.push_back(c.x);
            vtx_.push_back(c.y);
          }
        pvo_.push_back((uint32_t)(vtx_.size() / 2));
      }
      vpo_.push_back((uint32_t)pls_.size());
    }
    eg3d_scene& sc = scene_;
    sc.n_views = V;
    sc.width = sfmd.imageWidth_;
    sc.height = sfmd.imageHeight_;
    sc.cam_P = P_.data();
    sc.F = F_.data();
    sc.F_valid = Fv_.data();
    sc.view_pl_off = vpo_.data();
    sc.pl_vtx_off = pvo_.data();
    sc.vtx_xy = vtx_.data();
    sc.pl_start = pls_.data();
    sc.pl_end = ple_.data();
    sc.pl_valid = plv_.data();
    status_ = eg3d_create(&sc, device, &ctx_);
    if (status_ == EG3D_OK) upload_seeds();
  }
  ~PLGEdgeManager() {
    if (ctx_) eg3d_destroy(ctx_);
  }
  PLGEdgeManager(const PLGEdgeManager&) = delete;
  PLGEdgeManager& operator=(const PLGEdgeManager&) = delete;

  int last_status() const { return status_; }
  eg3d_ctx* ctx() const { return ctx_; }
  const eg3d_scene& scene() const { return scene_; }

  // pair(starting intersections, [per starting intersection][per track entry] correspondences), per track entry
  using per_view_result = std::pair<std::vector<PolyLineGraph2D::plg_point>,
                                    std::vector<std::vector<std::vector<PolyLineGraph2D::plg_point>>>>;
  std::vector<per_view_result> detect_nearby_intersections_and_correspondences_plgp(const int starting_point_id) {
    std::vector<per_view_result> res;
    if (!ctx_) return res;
    eg3d_seeds s = seeds_struct();
    eg3d_candidates c;
    status_ = eg3d_candidates_run(ctx_, &s, (uint32_t)starting_point_id, (uint32_t)starting_point_id + 1, &c);
    if (status_ != EG3D_OK) return res;
    const uint32_t k = c.n_sv;
    res.resize(k);
    for (uint32_t a = 0; a < k; a++)
      for (uint32_t h = c.start_off[a]; h < c.start_off[a + 1]; h++)
        res[a].first.push_back({c.start_pl[h], {c.start_seg[h], {c.start_xy[2 * h], c.start_xy[2 * h + 1]}}});
    for (uint32_t t = 0; t < c.n_tasks; t++) {
      std::vector<std::vector<PolyLineGraph2D::plg_point>> lists;
      for (uint32_t l = c.task_list_off[t]; l < c.task_list_off[t + 1]; l++) {
        std::vector<PolyLineGraph2D::plg_point> v;
        for (uint32_t h = c.list_off[l]; h < c.list_off[l + 1]; h++)
          v.push_back({c.hit_pl[h], {c.hit_seg[h], {c.hit_xy[2 * h], c.hit_xy[2 * h + 1]}}});
        lists.push_back(std::move(v));
      }
      res[c.task_sv[t]].second.push_back(std::move(lists));
    }
    eg3d_free_candidates(&c);
    return res;
  }

  // plg_matching_from_refpoints_parallel over all reference points. Seeds are independent
  // (plg_matching_from_refpoints.cpp:83-104), so the points are cut into batches that are kept in
  // flight on `contexts` clones of the context (eg3d_clone: shared scene and seeds, own stream), one
  // host thread each; batch results are appended in seed order, which is the reference's order.
  void set_batching(uint32_t seeds_per_batch, int contexts) {
    batch_ = seeds_per_batch ? seeds_per_batch : 2048;
    n_ctx_ = contexts < 1 ? 1 : contexts;
  }
  std::vector<new_3dpoint_plgp_matches> match_all(PLGMatchesManager* plgmm = nullptr) {
    std::vector<new_3dpoint_plgp_matches> res;
    if (!ctx_) return res;
    const uint32_t n = (uint32_t)sfmd_.numPoints_;
    const uint32_t nb = (n + batch_ - 1) / batch_;
    std::vector<eg3d_edgepoints> parts(nb);
    std::vector<int> rc(nb, EG3D_OK);
    std::vector<eg3d_ctx*> ctxs(1, ctx_);
    for (int k = 1; k < n_ctx_ && (uint32_t)k < nb; k++) {
      eg3d_ctx* c = nullptr;
      if (eg3d_clone(ctx_, &c) != EG3D_OK) break;
      ctxs.push_back(c);
    }
    std::atomic<uint32_t> next(0);
    auto work = [&](eg3d_ctx* c) {
      for (uint32_t i = next.fetch_add(1); i < nb; i = next.fetch_add(1)) {
        const uint32_t b = i * batch_, e = (b + batch_ < n) ? b + batch_ : n;
        rc[i] = eg3d_match_resident(c, b, e, 0, &parts[i], nullptr);
      }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < ctxs.size(); k++) th.emplace_back(work, ctxs[k]);
    work(ctxs[0]);
    for (auto& t : th) t.join();
    for (size_t k = 1; k < ctxs.size(); k++) eg3d_destroy(ctxs[k]);
    status_ = EG3D_OK;
    for (uint32_t i = 0; i < nb; i++)
      if (rc[i] != EG3D_OK) status_ = rc[i];
    if (status_ == EG3D_OK) {
      uint64_t total = 0;
      for (uint32_t i = 0; i < nb; i++) total += parts[i].n_points;
      res.reserve(total);
      for (uint32_t p = 0; p < nb; p++) {
        const eg3d_edgepoints& e = parts[p];
        for (uint64_t i = 0; i < e.n_points; i++) {
          std::vector<PolyLineGraph2D::plg_point> obs;
          std::vector<int> views;
          for (uint64_t j = e.obs_off[i]; j < e.obs_off[i + 1]; j++) {
            obs.push_back({e.obs_pl[j], {e.obs_seg[j], {e.obs_xy[2 * j], e.obs_xy[2 * j + 1]}}});
            views.push_back(e.obs_view[j]);
          }
          res.emplace_back(vec3{e.X[3 * i], e.X[3 * i + 1], e.X[3 * i + 2]}, std::move(obs), std::move(views));
        }
      }
      // plgmm.add_matched_3dpolyline(chain) for every emitted chain, in emission order
      // (plg_matching_from_refpoints.cpp:74-77): replayed on the host from the ordered cloud
      if (plgmm) status_ = plgmm->replay(scene_, parts);
    }
    for (uint32_t i = 0; i < nb; i++)
      if (rc[i] == EG3D_OK) eg3d_free_edgepoints(&parts[i]);
    return res;
  }

  // find_new_3d_points_from_compatible_polylines_expandallviews_parallel for ONE set of potentially
  // compatible polylines (per view: the reference's set<ulong> of polyline ids), as
  // pipelines.cpp:98,144 call it once per polyline match.
  std::vector<new_3dpoint_plgp_matches> match_polyline_set(const std::vector<std::set<unsigned long>>& compat) {
    std::vector<new_3dpoint_plgp_matches> res;
    if (!ctx_) return res;
    std::vector<uint32_t> row_off(1, 0), ids;
    for (const auto& per_view : compat) {
      for (unsigned long id : per_view) ids.push_back((uint32_t)id);
      row_off.push_back((uint32_t)ids.size());
    }
    if (ids.empty()) ids.push_back(0);
    eg3d_polyline_sets ps;
    ps.n_sets = 1;
    ps.row_off = row_off.data();
    ps.pl_ids = ids.data();
    eg3d_edgepoints e;
    status_ = eg3d_match_polyline_sets(ctx_, &ps, 0, 1, 0, &e, nullptr);
    if (status_ == EG3D_OK) {
      res.reserve(e.n_points);
      for (uint64_t i = 0; i < e.n_points; i++) {
        std::vector<PolyLineGraph2D::plg_point> obs;
        std::vector<int> views;
        for (uint64_t j = e.obs_off[i]; j < e.obs_off[i + 1]; j++) {
          obs.push_back({e.obs_pl[j], {e.obs_seg[j], {e.obs_xy[2 * j], e.obs_xy[2 * j + 1]}}});
          views.push_back(e.obs_view[j]);
        }
        res.emplace_back(vec3{e.X[3 * i], e.X[3 * i + 1], e.X[3 * i + 2]}, std::move(obs), std::move(views));
      }
      eg3d_free_edgepoints(&e);
    }
    return res;
  }

 private:
  eg3d_seeds seeds_struct() {
    eg3d_seeds s;
    s.n_seeds = (uint32_t)sfmd_.numPoints_;
    s.trk_off = toff_.data();
    s.trk_view = tview_.data();
    s.trk_xy = txy_.data();
    return s;
  }
  void upload_seeds() {
    toff_.assign(1, 0);
    for (int i = 0; i < sfmd_.numPoints_; i++) {
      for (size_t j = 0; j < sfmd_.camViewingPointN_[i].size(); j++) {
        tview_.push_back(sfmd_.camViewingPointN_[i][j]);
        txy_.push_back(sfmd_.point2DoncamViewingPoint_[i][j].x);
        txy_.push_back(sfmd_.point2DoncamViewingPoint_[i][j].y);
      }
      toff_.push_back((uint32_t)tview_.size());
    }
    eg3d_seeds s = seeds_struct();
    status_ = eg3d_upload_seeds(ctx_, &s);
  }
  const SfMData& sfmd_;
  std::vector<float> P_, vtx_, txy_;
  std::vector<double> F_;
  std::vector<uint8_t> Fv_, plv_;
  std::vector<uint32_t> vpo_, pvo_, pls_, ple_, toff_;
  std::vector<int32_t> tview_;
  eg3d_scene scene_;
  eg3d_ctx* ctx_ = nullptr;
  int status_ = EG3D_OK;
  uint32_t batch_ = 2048;
  int n_ctx_ = 3;
};

// plg_matching_from_refpoints[_parallel]: the consensus manager and the matches manager of the
// reference signature are folded into the edge manager / replayable from the ordered output.
inline std::vector<new_3dpoint_plgp_matches> plg_matching_from_refpoints_parallel(const SfMData&, PLGEdgeManager* em) {
  return em->match_all();
}
// ... and with the matches manager of the reference signature (sfmd, em, cm, plgmm): the consensus
// manager lives in the edge manager; plgmm receives what add_matched_3dpolyline would have recorded
inline std::vector<new_3dpoint_plgp_matches> plg_matching_from_refpoints_parallel(const SfMData&, PLGEdgeManager* em,
                                                                                  PLGMatchesManager& plgmm) {
  return em->match_all(&plgmm);
}
inline std::vector<new_3dpoint_plgp_matches> plg_matching_from_refpoints(const SfMData& s, PLGEdgeManager* em) {
  return plg_matching_from_refpoints_parallel(s, em);
}

// find_new_3d_points_from_compatible_polylines_expandallviews[_parallel]
// (include/edgegraph3d/matching/plg_matching/polyline_matching.hpp:55-56): plgs, F, the matches
// manager and the grid maps of the reference signature live in the edge manager.
inline std::vector<new_3dpoint_plgp_matches> find_new_3d_points_from_compatible_polylines_expandallviews_parallel(
    const SfMData&, PLGEdgeManager* em, const std::vector<std::set<unsigned long>>& potentially_compatible_polylines) {
  return em->match_polyline_set(potentially_compatible_polylines);
}

// gaussNewtonFiltering(SfMData&, vector<bool>&, float): mutates points_ of the inliers in place.
inline void gaussNewtonFiltering(SfMData& sfm_data_, std::vector<bool>& inliers, const float gn_max_mse,
                                 PLGEdgeManager* em) {
  const size_t n = sfm_data_.points_.size();
  inliers.assign(n, false);
  std::vector<float> X(3 * n), Xo(3 * n), xy;
  std::vector<uint32_t> off(1, 0);
  std::vector<int32_t> view;
  for (size_t i = 0; i < n; i++) {
    X[3 * i] = sfm_data_.points_[i].x;
    X[3 * i + 1] = sfm_data_.points_[i].y;
    X[3 * i + 2] = sfm_data_.points_[i].z;
    for (size_t j = 0; j < sfm_data_.camViewingPointN_[i].size(); j++) {
      view.push_back(sfm_data_.camViewingPointN_[i][j]);
      xy.push_back(sfm_data_.point2DoncamViewingPoint_[i][j].x);
      xy.push_back(sfm_data_.point2DoncamViewingPoint_[i][j].y);
    }
    off.push_back((uint32_t)view.size());
  }
  std::vector<uint8_t> inl(n ? n : 1);
  if (eg3d_gn_filter(em->ctx(), X.data(), off.data(), view.data(), xy.data(), n, gn_max_mse, 0, Xo.data(), inl.data(),
                     nullptr) != EG3D_OK)
    return;
  for (size_t i = 0; i < n; i++)
    if (inl[i]) {
      sfm_data_.points_[i] = {Xo[3 * i], Xo[3 * i + 1], Xo[3 * i + 2]};
      inliers[i] = true;
    }
}

}  // namespace eg3d_ref
