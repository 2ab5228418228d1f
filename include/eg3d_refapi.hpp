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
//   filter(SfMData&, int[, float][, int])   (4 overloads)
//        include/edgegraph3d/filtering/outliers_filtering.hpp:18-21
//   EdgeManager (abstract), PLGPConsensusManager (abstract), PLGPCM3ViewsPLGFollowing
//        include/edgegraph3d/edge_managers/edge_manager.hpp:54-73,
//        include/edgegraph3d/matching/consensus_manager/plgp_consensus_manager.hpp:56-72, plgpcm_3views_plg_following.hpp:52-60
// so that the reference's call sites compile unchanged against these types:
//   pipelines.cpp:164        plg_matching_from_refpoints_parallel(sfmd, em, cm, plgmm)
//   outliers_filtering.cpp:39 gaussNewtonFiltering(sfm_data_, inliers, gn_max_mse)
//   edge_matcher.cpp:132      filter(sfmd, first_edgepoint)
// (include/eg3d_refapi_glm.hpp converts the reference's glm-typed structures to and from the ones below.)
//
// Types: the structs below have the fields of the reference types the path reads
// (SfMData.h:16-30, types_reconstructor.hpp:68-82, polyline_graph_2d.hpp:85-119,222-294) with
// plain float arrays in place of glm/cv::Mat. Error behaviour: the reference's functions have no error channel
// (an empty result means "no edges found"), so a FAILURE of the GPU path must not look like one — the functions with
// the reference's exact signatures throw eg3d_ref::Eg3dError (status + eg3d_last_error() text); the manager's own
// methods (match_all, match_polyline_set, detect_...) return an empty result and keep the status in last_status().
#pragma once
#include <array>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <set>
#include <stdexcept>
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
struct SfMData {  // SfMData.h:16-30
  int numPoints_ = 0, numCameras_ = 0;
  std::vector<vec3> points_;
  std::vector<CameraType> camerasList_;
  std::vector<std::string> camerasPaths_;
  std::vector<std::vector<int>> camViewingPointN_;
  std::vector<std::vector<int>> pointsVisibleFromCamN_;
  std::vector<std::vector<vec2>> point2DoncamViewingPoint_;
  int imageWidth_ = 0, imageHeight_ = 0;
};
// A failure of the GPU path behind a function that has the reference's signature (and therefore no status to return)
struct Eg3dError : std::runtime_error {
  int status;
  Eg3dError(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};
// thrown by the legacy (segment-based) virtuals of EdgeManager, as the reference's PLGEdgeManager does
// (plg_edge_manager.cpp:361-378: `throw new NotImplementedException()`)
struct NotImplementedException : std::logic_error {
  NotImplementedException() : std::logic_error("Function not yet implemented") {}
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

  // the same from chains held as the reference holds them (vector of point tuples per chain; key = seed, index of the
  // chain within the seed): every chain in the order given = the order add_matched_3dpolyline would have seen
  int replay_chains(const eg3d_scene& sc, const std::vector<std::vector<new_3dpoint_plgp_matches>>& chains,
                    const std::vector<std::array<uint32_t, 3>>& chain_key) {
    eg3d_host_free_graph3d(&g_);
    std::vector<float> X, xy;
    std::vector<uint32_t> pl, seg, key;
    std::vector<uint64_t> off(1, 0);
    std::vector<int32_t> view;
    for (size_t c = 0; c < chains.size(); c++)
      for (size_t i = 0; i < chains[c].size(); i++) {
        const vec3& p = std::get<0>(chains[c][i]);
        X.insert(X.end(), {p.x, p.y, p.z});
        const uint32_t k[4] = {chain_key[c][0], chain_key[c][1], chain_key[c][2], (uint32_t)i};
        key.insert(key.end(), k, k + 4);
        const auto& obs = std::get<1>(chains[c][i]);
        const auto& vs = std::get<2>(chains[c][i]);
        for (size_t j = 0; j < obs.size(); j++) {
          view.push_back(vs[j]);
          pl.push_back((uint32_t)obs[j].polyline_id);
          seg.push_back((uint32_t)obs[j].plp.segment_index);
          xy.push_back(obs[j].plp.coords.x);
          xy.push_back(obs[j].plp.coords.y);
        }
        off.push_back((uint64_t)view.size());
      }
    if (view.empty()) {  // nothing matched: an empty graph, not an error
      view.push_back(0);
      pl.push_back(0);
      seg.push_back(0);
      xy.assign(2, 0.f);
    }
    if (X.empty()) {
      X.assign(3, 0.f);
      key.assign(4, 0);
    }
    eg3d_edgepoints all;
    std::memset(&all, 0, sizeof(all));
    all.n_points = off.size() - 1;
    all.n_obs = all.n_points ? off.back() : 0;
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

// EdgeManager (edge_manager.hpp:54-73): the abstract base the path's entry points take (`const EdgeManager* em`, which
// plg_matching_from_refpoints.cpp:67 downcasts to PLGEdgeManager*). The cv::Mat image accessors of the reference's
// base are not part of this path and are left out; the two pure virtuals keep their signatures.
class EdgeManager {
 public:
  virtual std::vector<vec2> detect_nearby_edge_intersections(const int imgId, const int startingPoint_id,
                                                             const float starting_detection_dist) = 0;
  virtual std::vector<std::vector<vec2>> detect_epipolar_intersections(const int starting_image_id, const int starting_point_id,
                                                                       const vec2& intersection_point_on_starting_image,
                                                                       const float max_correspondence_detection_radius) = 0;
  virtual ~EdgeManager() {}

 protected:
  explicit EdgeManager(const SfMData& input_sfmd) : sfmd(input_sfmd) {}
  const SfMData& sfmd;
};

// Owns the flattened scene and the GPU context: what `new PLGEdgeManager(imgs, sfmd, F, plgs, 10, 3)` and the 4 px
// PolyLine2DMapSearch maps hold in the reference (edge_matcher.cpp:101-107), resident in HBM.
class PLGEdgeManager : public EdgeManager {
 public:
  // the reference's constructor shape minus the images (plg_edge_manager.hpp:80): the two detection constants are
  // compile-time constants of the kernels (global_defines.hpp:35-36), other values are refused
  PLGEdgeManager(const SfMData& sfmd, const FundamentalMatrices& F, const std::vector<PolyLineGraph2D>& plgs,
                 const float starting_detection_dist, const float correspondence_detection_range_multiplication_factor,
                 int device = 0)
      : PLGEdgeManager(sfmd, F, plgs, device) {
    if (starting_detection_dist != 10.0f || correspondence_detection_range_multiplication_factor != 3.0f)
      throw std::invalid_argument("eg3d_ref::PLGEdgeManager: DETECTION_STARTING_RADIUS 10 / MULTIPLICATION_FACTOR 3 are built in");
  }
  // legacy segment-based interface: not implemented by the reference's PLGEdgeManager either
  std::vector<vec2> detect_nearby_edge_intersections(const int, const int, const float) override { throw NotImplementedException(); }
  std::vector<std::vector<vec2>> detect_epipolar_intersections(const int, const int, const vec2&, const float) override {
    throw NotImplementedException();
  }
  // the manager the 3-argument gaussNewtonFiltering / filter() take their GPU context from: the most recently
  // constructed one that is still alive (single caller thread, as in the reference)
  static PLGEdgeManager*& default_manager() {
    static PLGEdgeManager* d = nullptr;
    return d;
  }
  PLGEdgeManager(const SfMData& sfmd, const FundamentalMatrices& F, const std::vector<PolyLineGraph2D>& plgs,
                 int device = 0)
      : EdgeManager(sfmd), sfmd_(sfmd) {
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
    if (status_ == EG3D_OK) default_manager() = this;
  }
  ~PLGEdgeManager() override {
    if (default_manager() == this) default_manager() = nullptr;
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

  // plg_matching_from_refpoints_parallel over all reference points: ONE call of the C ABI. Seeds are independent
  // (plg_matching_from_refpoints.cpp:83-104 runs them on its OpenMP team); since round 6 the library itself cuts a call
  // into sub-batches that it keeps in flight on internal contexts and concatenates in seed order, which is the
  // reference's order (eg3d_set_pipelining; until round 5 this shim did the same above the ABI with eg3d_clone).
  // set_batching(seeds per sub-batch, sub-batches in flight): 0 / 0 = the library's defaults.
  void set_batching(uint32_t seeds_per_batch, int contexts) {
    batch_ = seeds_per_batch;
    n_ctx_ = contexts < 0 ? 0 : contexts;
  }
  std::vector<new_3dpoint_plgp_matches> match_all(PLGMatchesManager* plgmm = nullptr) {
    std::vector<new_3dpoint_plgp_matches> res;
    if (!ctx_) return res;
    const uint32_t n = (uint32_t)sfmd_.numPoints_;
    const uint32_t nb = 1;
    std::vector<eg3d_edgepoints> parts(nb);
    std::vector<int> rc(nb, EG3D_OK);
    eg3d_set_pipelining(ctx_, n_ctx_, batch_ ? (int)((n + batch_ - 1) / batch_) : 0);
    rc[0] = eg3d_match_resident(ctx_, 0, n, 0, &parts[0], nullptr);
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

  // All chains of ONE reference point, grouped the way the reference's loop produces them
  // (plg_matching_from_refpoints.cpp:69-78): [track entry][starting intersection] -> chain (possibly empty).
  // n_start[entry] = starting intersections of that entry (a chain-less intersection still has its — empty — slot).
  using chain = std::vector<new_3dpoint_plgp_matches>;
  std::vector<std::vector<chain>> match_refpoint_chains(const unsigned long refpoint_id) {
    std::vector<std::vector<chain>> res;
    if (!ctx_ || refpoint_id >= (unsigned long)sfmd_.numPoints_) return res;
    eg3d_seeds s = seeds_struct();
    eg3d_candidates c;
    status_ = eg3d_candidates_run(ctx_, &s, (uint32_t)refpoint_id, (uint32_t)refpoint_id + 1, &c);
    if (status_ != EG3D_OK) return res;
    res.resize(c.n_sv);
    for (uint32_t a = 0; a < c.n_sv; a++) res[a].resize(c.start_off[a + 1] - c.start_off[a]);
    eg3d_free_candidates(&c);
    eg3d_edgepoints e;
    status_ = eg3d_match_resident(ctx_, (uint32_t)refpoint_id, (uint32_t)refpoint_id + 1, 0, &e, nullptr);
    if (status_ != EG3D_OK) {
      res.clear();
      return res;
    }
    for (uint64_t i = 0; i < e.n_points; i++) {
      const uint32_t entry = e.key[4 * i + 1], hit = e.key[4 * i + 2];
      if (entry >= res.size() || hit >= res[entry].size()) continue;
      std::vector<PolyLineGraph2D::plg_point> obs;
      std::vector<int> views;
      for (uint64_t j = e.obs_off[i]; j < e.obs_off[i + 1]; j++) {
        obs.push_back({e.obs_pl[j], {e.obs_seg[j], {e.obs_xy[2 * j], e.obs_xy[2 * j + 1]}}});
        views.push_back(e.obs_view[j]);
      }
      res[entry][hit].emplace_back(vec3{e.X[3 * i], e.X[3 * i + 1], e.X[3 * i + 2]}, std::move(obs), std::move(views));
    }
    eg3d_free_edgepoints(&e);
    return res;
  }
  const SfMData& sfm_data() const { return sfmd_; }

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
  uint32_t batch_ = 0;
  int n_ctx_ = 0;
};

// PLGPConsensusManager (plgp_consensus_manager.hpp:56-72): the strategy interface of the path, same two pure virtuals.
class PLGPConsensusManager {
 public:
  using points = std::vector<new_3dpoint_plgp_matches>;
  using intersections_and_correspondences =
      std::pair<std::vector<PolyLineGraph2D::plg_point>, std::vector<std::vector<std::vector<PolyLineGraph2D::plg_point>>>>;
  virtual points consensus_strategy_single_point(const int starting_img_id, const int refpoint,
                                                 const intersections_and_correspondences& intersections_and_correspondences_pair) = 0;
  virtual std::vector<points> consensus_strategy_single_point_vector(
      const int starting_img_id, const int refpoint, const intersections_and_correspondences& intersections_and_correspondences_pair) = 0;
  virtual ~PLGPConsensusManager() {}
};

// PLGPCM3ViewsPLGFollowing (plgpcm_3views_plg_following.hpp:52-60; .cpp:40-69): 3-view consensus with polyline
// following and expand-to-all-views — kernels task_setup, K3a, K3s, K3b of the edge manager's context. The
// intersections handed in are the ones the SAME edge manager produced for that reference point (the only way the
// reference calls it, plg_matching_from_refpoints.cpp:67-72); the GPU path recomputes them, so their content is only
// used to tell apart two track entries of the same view (Q2).
class PLGPCM3ViewsPLGFollowing : public PLGPConsensusManager {
 public:
  explicit PLGPCM3ViewsPLGFollowing(PLGEdgeManager& em) : em_(em) {}
  PLGEdgeManager& edge_manager() const { return em_; }
  std::vector<points> consensus_strategy_single_point_vector(
      const int starting_img_id, const int refpoint, const intersections_and_correspondences& pair) override {
    if (cached_point_ != refpoint) {
      cache_ = em_.match_refpoint_chains((unsigned long)refpoint);
      if (em_.last_status() != EG3D_OK) throw Eg3dError(em_.last_status(), std::string("eg3d: ") + eg3d_last_error());
      cached_point_ = refpoint;
      served_.assign(cache_.size(), 0);
    }
    // The track entry this call is about. The reference's loop (plg_matching_from_refpoints.cpp:68-73) calls once per
    // track entry, in track order, so a view id that appears on several entries of the track is served entry by entry:
    // the first entry of that view not handed out yet (all handed out: the round starts again — a caller that walks
    // the track a second time). An entry whose starting intersections differ in number from the ones handed in is not
    // the caller's entry and is skipped while another candidate remains.
    const std::vector<int>& track = em_.sfm_data().camViewingPointN_[refpoint];
    int entry = -1, fallback = -1;
    for (int round = 0; round < 2 && entry < 0; round++) {
      for (size_t i = 0; i < track.size() && i < cache_.size(); i++) {
        if (track[i] != starting_img_id || served_[i]) continue;
        if (fallback < 0) fallback = (int)i;
        if (cache_[i].size() == pair.first.size()) {
          entry = (int)i;
          break;
        }
      }
      if (entry < 0 && fallback >= 0) entry = fallback;
      if (entry < 0)  // every entry of this view was served: forget and look again
        for (size_t i = 0; i < track.size() && i < served_.size(); i++)
          if (track[i] == starting_img_id) served_[i] = 0;
    }
    if (entry >= 0) served_[(size_t)entry] = 1;
    std::vector<points> res(pair.first.size());
    if (entry >= 0)
      for (size_t h = 0; h < res.size() && h < cache_[entry].size(); h++) res[h] = cache_[entry][h];
    return res;
  }
  points consensus_strategy_single_point(const int starting_img_id, const int refpoint,
                                         const intersections_and_correspondences& pair) override {
    points res;
    for (auto& ch : consensus_strategy_single_point_vector(starting_img_id, refpoint, pair))
      for (auto& p : ch) res.push_back(std::move(p));
    return res;
  }

 private:
  PLGEdgeManager& em_;
  int cached_point_ = -1;
  std::vector<std::vector<PLGEdgeManager::chain>> cache_;
  std::vector<uint8_t> served_;  // per track entry of the cached point: its chains were handed out
};

namespace detail {
inline void throw_if_failed(const PLGEdgeManager* em, const char* what) {
  if (em->last_status() != EG3D_OK)
    throw Eg3dError(em->last_status(), std::string(what) + ": " + eg3d_last_error());
}
}  // namespace detail

// plg_matching_from_refpoints[_parallel](sfm_data, em, cm, plgmm) — plg_matching_from_refpoints.hpp:53,55, the call of
// pipelines.cpp:164. `em` is downcast to PLGEdgeManager* exactly as plg_matching_from_refpoints.cpp:67 does. With the
// consensus manager of the path (PLGPCM3ViewsPLGFollowing of the same edge manager, what edge_matcher.cpp:115 builds)
// all reference points run as batches in flight on the GPU; any other PLGPConsensusManager is honoured through the
// reference's own per-point loop (candidates from the GPU, consensus from the caller's strategy). plgmm receives what
// add_matched_3dpolyline(chain) would have recorded for every chain, in emission order. Throws Eg3dError when the GPU
// path fails (the reference's signature has no status).
inline std::vector<new_3dpoint_plgp_matches> plg_matching_from_refpoint(const SfMData& sfm_data, const EdgeManager* em,
                                                                        const PLGPConsensusManager* cm,
                                                                        const unsigned long refpoint_id,
                                                                        std::vector<PLGEdgeManager::chain>* chains_out = nullptr) {
  std::vector<new_3dpoint_plgp_matches> res;
  PLGEdgeManager* g = (PLGEdgeManager*)em;
  PLGPConsensusManager* c = const_cast<PLGPConsensusManager*>(cm);  // the reference calls the non-const virtual through its const pointer (-fpermissive)
  auto all_imgs = g->detect_nearby_intersections_and_correspondences_plgp((int)refpoint_id);
  detail::throw_if_failed(g, "detect_nearby_intersections_and_correspondences_plgp");
  for (size_t i = 0; i < sfm_data.camViewingPointN_[refpoint_id].size() && i < all_imgs.size(); i++) {
    const int starting_img_id = sfm_data.camViewingPointN_[refpoint_id][i];
    auto cur_res_vec = c->consensus_strategy_single_point_vector(starting_img_id, (int)refpoint_id, all_imgs[i]);
    for (auto& cur_res : cur_res_vec) {
      res.insert(res.end(), cur_res.begin(), cur_res.end());
      if (chains_out) chains_out->push_back(std::move(cur_res));
    }
  }
  return res;
}
inline std::vector<new_3dpoint_plgp_matches> plg_matching_from_refpoints_parallel(const SfMData& sfm_data, const EdgeManager* em,
                                                                                  const PLGPConsensusManager* cm,
                                                                                  PLGMatchesManager& plgmm) {
  PLGEdgeManager* g = (PLGEdgeManager*)em;
  const PLGPCM3ViewsPLGFollowing* own = dynamic_cast<const PLGPCM3ViewsPLGFollowing*>(cm);
  if (own && &own->edge_manager() == g) {
    auto res = g->match_all(&plgmm);
    detail::throw_if_failed(g, "plg_matching_from_refpoints_parallel");
    return res;
  }
  // a caller-supplied consensus strategy: the reference's loop, point by point
  std::vector<new_3dpoint_plgp_matches> res;
  std::vector<PLGEdgeManager::chain> chains;
  std::vector<std::array<uint32_t, 3>> chain_key;
  for (unsigned long r = 0; r < (unsigned long)sfm_data.numPoints_; r++) {
    const size_t before = chains.size();
    auto cur = plg_matching_from_refpoint(sfm_data, em, cm, r, &chains);
    res.insert(res.end(), cur.begin(), cur.end());
    for (size_t k = before; k < chains.size(); k++) chain_key.push_back({(uint32_t)r, (uint32_t)(k - before), 0u});
  }
  const int rc = plgmm.replay_chains(g->scene(), chains, chain_key);
  if (rc != 0) throw Eg3dError(rc, "eg3d_host_replay_matches failed");
  return res;
}
inline std::vector<new_3dpoint_plgp_matches> plg_matching_from_refpoints(const SfMData& sfm_data, const EdgeManager* em,
                                                                         const PLGPConsensusManager* cm, PLGMatchesManager& plgmm) {
  return plg_matching_from_refpoints_parallel(sfm_data, em, cm, plgmm);  // the reference's "parallel" loop is serial too
}

// Shorter forms kept from earlier rounds (the consensus manager of the path implied): empty result + last_status() on failure.
inline std::vector<new_3dpoint_plgp_matches> plg_matching_from_refpoints_parallel(const SfMData&, PLGEdgeManager* em) {
  return em->match_all();
}
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

namespace detail {
// A GPU context for the filter: the registered edge manager's when it serves the same rig, else a private one built
// from the cameras alone (a scene without polylines — ./filter runs without any polyline graph, filter.cpp:48-115).
struct FilterContext {
  eg3d_ctx* ctx = nullptr;
  bool own = false;
  explicit FilterContext(const SfMData& s) {
    const int V = s.numCameras_;
    if (V <= 0 || s.camerasList_.size() < (size_t)V)
      throw Eg3dError(EG3D_ERR_ARG, "filter: the SfM data holds no cameras (numCameras_ / camerasList_)");
    // the registered manager's context serves this rig only when EVERY camera matrix is the same (a different rig that
    // shares camera 0 must not be filtered with the manager's cameras)
    PLGEdgeManager* d = PLGEdgeManager::default_manager();
    if (d && d->ctx() && d->scene().n_views == V) {
      bool same = true;
      for (int v = 0; v < V && same; v++)
        same = std::memcmp(d->scene().cam_P + (size_t)v * 16, &s.camerasList_[(size_t)v].cameraMatrix[0][0], sizeof(float) * 16) == 0;
      if (same) {
        ctx = d->ctx();
        return;
      }
    }
    std::vector<float> P;
    for (int v = 0; v < V; v++)
      for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) P.push_back(s.camerasList_[v].cameraMatrix[r][c]);
    std::vector<double> F((size_t)V * V * 9, 0.0);
    std::vector<uint8_t> Fv((size_t)V * V, 0);
    std::vector<uint32_t> vpo((size_t)V + 1, 0), pvo(1, 0), none(1, 0);
    std::vector<float> vtx(2, 0.f);
    std::vector<uint8_t> plv(1, 0);
    eg3d_scene sc;
    std::memset(&sc, 0, sizeof(sc));
    sc.n_views = V;
    sc.width = s.imageWidth_ > 0 ? s.imageWidth_ : 1;
    sc.height = s.imageHeight_ > 0 ? s.imageHeight_ : 1;
    sc.cam_P = P.data();
    sc.F = F.data();
    sc.F_valid = Fv.data();
    sc.view_pl_off = vpo.data();
    sc.pl_vtx_off = pvo.data();
    sc.vtx_xy = vtx.data();
    sc.pl_start = none.data();
    sc.pl_end = none.data();
    sc.pl_valid = plv.data();
    const int rc = eg3d_create(&sc, 0, &ctx);
    if (rc != EG3D_OK) throw Eg3dError(rc, std::string("eg3d_create (filter context): ") + eg3d_last_error());
    own = true;
  }
  ~FilterContext() {
    if (own && ctx) eg3d_destroy(ctx);
  }
  FilterContext(const FilterContext&) = delete;
  FilterContext& operator=(const FilterContext&) = delete;
};
}  // namespace detail

// gaussNewtonFiltering(SfMData&, std::vector<bool>&, const float) — gauss_newton.hpp:20 (gauss_newton.cpp:136-178):
// FP32 Gauss-Newton from the stored X over all observations of every point; inliers get their point replaced. The GPU
// context is the registered edge manager's (PLGEdgeManager::default_manager()) or a private one. Throws Eg3dError on
// a failure of the GPU path.
inline void gaussNewtonFiltering(SfMData& sfm_data_, std::vector<bool>& inliers, const float gn_max_mse) {
  const size_t n = sfm_data_.points_.size();
  inliers.assign(n, false);
  if (!n) return;
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
    if (view.size() > 0xffffffffull) throw Eg3dError(EG3D_ERR_ARG, "gaussNewtonFiltering: more than 2^32 - 1 observations");
    off.push_back((uint32_t)view.size());
  }
  if (view.empty()) {
    view.push_back(0);
    xy.assign(2, 0.f);
  }
  std::vector<uint8_t> inl(n);
  detail::FilterContext fc(sfm_data_);
  const int rc = eg3d_gn_filter(fc.ctx, X.data(), off.data(), view.data(), xy.data(), n, gn_max_mse, 0, Xo.data(), inl.data(), nullptr);
  if (rc != EG3D_OK) throw Eg3dError(rc, std::string("eg3d_gn_filter: ") + eg3d_last_error());
  for (size_t i = 0; i < n; i++)
    if (inl[i]) {
      sfm_data_.points_[i] = {Xo[3 * i], Xo[3 * i + 1], Xo[3 * i + 2]};
      inliers[i] = true;
    }
}
// earlier rounds' form with an explicit manager (kept): the manager's context, same behaviour
inline void gaussNewtonFiltering(SfMData& sfm_data_, std::vector<bool>& inliers, const float gn_max_mse, PLGEdgeManager* em) {
  PLGEdgeManager*& d = PLGEdgeManager::default_manager();
  PLGEdgeManager* saved = d;
  d = em;
  try {
    gaussNewtonFiltering(sfm_data_, inliers, gn_max_mse);
  } catch (...) {
    d = saved;
    throw;
  }
  d = saved;
}

#define EG3D_REF_GN_MAX_MSE 2.25f           /* GN_MAX_MSE, gauss_newton.hpp:18 */
#define EG3D_REF_INVALID_FORCED_MIN_FILTER -1 /* outliers_filtering.cpp:12 */

// compute_inliers (outliers_filtering.cpp:37-64): Gauss-Newton inliers, then points >= first_edgepoint need MORE than
// max(3, median track length / 2 - 1) observations (or the forced amount)
inline std::vector<bool> compute_inliers(SfMData& sfm_data_, const int first_edgepoint, const float gn_max_mse,
                                         const int forced_min_filter) {
  std::vector<bool> inliers;
  gaussNewtonFiltering(sfm_data_, inliers, gn_max_mse);
  const size_t n = sfm_data_.points_.size();
  std::vector<uint32_t> off(1, 0);
  for (size_t i = 0; i < n; i++) off.push_back(off.back() + (uint32_t)sfm_data_.camViewingPointN_[i].size());
  std::vector<uint8_t> inl(n ? n : 1);
  for (size_t i = 0; i < n; i++) inl[i] = inliers[i] ? 1 : 0;
  eg3d_host_observation_filter(sfm_data_.numCameras_, off.data(), n, (uint64_t)(first_edgepoint < 0 ? 0 : first_edgepoint),
                               forced_min_filter, inl.data());
  for (size_t i = 0; i < n; i++) inliers[i] = inl[i] != 0;
  return inliers;
}
// removeOutliers (outliers_filtering.cpp:66-92)
inline void removeOutliers(SfMData& sfmd, const std::vector<bool>& inliers) {
  SfMData res;
  res.camerasList_ = sfmd.camerasList_;
  res.camerasPaths_ = sfmd.camerasPaths_;
  res.numCameras_ = sfmd.numCameras_;
  res.imageWidth_ = sfmd.imageWidth_;
  res.imageHeight_ = sfmd.imageHeight_;
  res.pointsVisibleFromCamN_.resize(res.numCameras_);
  int curpt = 0;
  for (size_t i = 0; i < inliers.size(); i++)
    if (inliers[i]) {
      res.points_.push_back(sfmd.points_[i]);
      res.camViewingPointN_.push_back(sfmd.camViewingPointN_[i]);
      res.point2DoncamViewingPoint_.push_back(sfmd.point2DoncamViewingPoint_[i]);
      for (const int cam_id : sfmd.camViewingPointN_[i]) res.pointsVisibleFromCamN_[cam_id].push_back(curpt);
      curpt++;
    }
  res.numPoints_ = (int)res.points_.size();
  sfmd = res;
}
// filter(): the four overloads of outliers_filtering.hpp:18-21 (outliers_filtering.cpp:94-114), without the prints
inline void filter(SfMData& sfmd, const int first_edgepoint, const float gn_max_mse, const int forced_min_filter) {
  const std::vector<bool> inliers = compute_inliers(sfmd, first_edgepoint, gn_max_mse, forced_min_filter);
  removeOutliers(sfmd, inliers);
}
inline void filter(SfMData& sfmd, const int first_edgepoint) {
  filter(sfmd, first_edgepoint, EG3D_REF_GN_MAX_MSE, EG3D_REF_INVALID_FORCED_MIN_FILTER);
}
inline void filter(SfMData& sfmd, const int first_edgepoint, const float gn_max_mse) {
  filter(sfmd, first_edgepoint, gn_max_mse, EG3D_REF_INVALID_FORCED_MIN_FILTER);
}
inline void filter(SfMData& sfmd, const int first_edgepoint, const int forced_min_filter) {
  filter(sfmd, first_edgepoint, EG3D_REF_GN_MAX_MSE, forced_min_filter);
}

}  // namespace eg3d_ref
