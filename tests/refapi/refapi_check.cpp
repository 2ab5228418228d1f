// GPU check of include/eg3d_refapi.hpp (the reference's call surface over the C ABI): a synthetic
// scene is poured into the shim's SfMData / PolyLineGraph2D / F types the way reference-side
// host code holds them, matched through PLGEdgeManager::match_all() with several batches in
// flight, and compared with one direct eg3d_match_refpoints call on the same scene.
// Build: g++ -std=c++17 -pthread -I include refapi_check.cpp -L edgegraph3d_amd -leg3d -leg3d_host ...
#include <cstdio>
#include <cstring>

#include "eg3d_host.h"
#include "eg3d_refapi.hpp"

using namespace eg3d_ref;

int main(int argc, char** argv) {
  const int cfg_index = argc > 1 ? atoi(argv[1]) : 1;
  eg3d_synth_config cfg;
  eg3d_synth_default_config(&cfg, cfg_index);
  eg3d_synth* syn = eg3d_synth_create(&cfg);
  const eg3d_scene* sc = eg3d_synth_scene(syn);
  const eg3d_seeds* sd = eg3d_synth_seeds(syn);
  const int V = sc->n_views;

  SfMData sfm;
  sfm.numCameras_ = V;
  sfm.numPoints_ = (int)sd->n_seeds;
  sfm.imageWidth_ = sc->width;
  sfm.imageHeight_ = sc->height;
  sfm.camerasList_.resize(V);
  for (int v = 0; v < V; v++)
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) sfm.camerasList_[v].cameraMatrix[r][c] = sc->cam_P[v * 16 + r * 4 + c];
  sfm.points_.assign(sd->n_seeds, vec3{0, 0, 0});
  sfm.camViewingPointN_.resize(sd->n_seeds);
  sfm.point2DoncamViewingPoint_.resize(sd->n_seeds);
  for (uint32_t i = 0; i < sd->n_seeds; i++)
    for (uint32_t j = sd->trk_off[i]; j < sd->trk_off[i + 1]; j++) {
      sfm.camViewingPointN_[i].push_back(sd->trk_view[j]);
      sfm.point2DoncamViewingPoint_[i].push_back(vec2{sd->trk_xy[2 * j], sd->trk_xy[2 * j + 1]});
    }
  FundamentalMatrices F(V, std::vector<std::array<double, 9>>(V));
  for (int i = 0; i < V; i++)
    for (int j = 0; j < V; j++)
      for (int k = 0; k < 9; k++) F[i][j][k] = sc->F_valid[i * V + j] ? sc->F[((size_t)i * V + j) * 9 + k] : 0.0;
  // polyline graphs: node ids of the flat scene become node coordinates = the end vertices
  std::vector<PolyLineGraph2D> plgs(V);
  for (int v = 0; v < V; v++) {
    PolyLineGraph2D& g = plgs[v];
    for (uint32_t p = sc->view_pl_off[v]; p < sc->view_pl_off[v + 1]; p++) {
      PolyLineGraph2D::polyline pl;
      const uint32_t a = sc->pl_vtx_off[p], b = sc->pl_vtx_off[p + 1];
      for (uint32_t k = a; k < b; k++) pl.polyline_coords.push_back(vec2{sc->vtx_xy[2 * k], sc->vtx_xy[2 * k + 1]});
      pl.start = sc->pl_start[p];
      pl.end = sc->pl_end[p];
      const unsigned long hi = pl.start > pl.end ? pl.start : pl.end;
      if (g.nodes_coords.size() <= hi) g.nodes_coords.resize(hi + 1, vec2{-1, -1});
      if (b - a > 1 && sc->pl_valid[p]) {
        g.nodes_coords[pl.start] = pl.polyline_coords.front();
        g.nodes_coords[pl.end] = pl.polyline_coords.back();
      }
      g.polylines.push_back(std::move(pl));
    }
  }

  PLGEdgeManager em(sfm, F, plgs, 0);
  if (em.last_status() != EG3D_OK) {
    std::printf("FAIL create: %s\n", eg3d_last_error());
    return 1;
  }
  em.set_batching(37, 3);  // ragged batches, three contexts in flight
  PLGMatchesManager plgmm;   // row a17: filled by the replay of the batched run
  auto many = plg_matching_from_refpoints_parallel(sfm, &em, plgmm);
  if (em.last_status() != EG3D_OK) {
    std::printf("FAIL match_all: %s\n", eg3d_last_error());
    return 1;
  }
  em.set_batching(1u << 30, 1);  // one batch, one context
  auto one = plg_matching_from_refpoints(sfm, &em);

  // the same scene straight through the C ABI
  eg3d_ctx* ctx = nullptr;
  eg3d_edgepoints e;
  if (eg3d_create(sc, 0, &ctx) != EG3D_OK || eg3d_match_refpoints(ctx, sd, 0, sd->n_seeds, 0, &e, nullptr) != EG3D_OK) {
    std::printf("FAIL direct: %s\n", eg3d_last_error());
    return 1;
  }
  int bad = 0;
  if (many.size() != e.n_points || one.size() != e.n_points) bad++;
  for (uint64_t i = 0; i < e.n_points && !bad; i++) {
    for (const auto* r : {&many[i], &one[i]}) {
      const vec3& X = std::get<0>(*r);
      if (std::memcmp(&X, e.X + 3 * i, 12) != 0) bad++;
      const auto& obs = std::get<1>(*r);
      const auto& views = std::get<2>(*r);
      if (obs.size() != e.obs_off[i + 1] - e.obs_off[i]) {
        bad++;
        continue;
      }
      for (size_t j = 0; j < obs.size(); j++) {
        const uint64_t o = e.obs_off[i] + (uint64_t)j;
        if (views[j] != e.obs_view[o] || obs[j].polyline_id != e.obs_pl[o] || obs[j].plp.segment_index != e.obs_seg[o] ||
            std::memcmp(&obs[j].plp.coords, e.obs_xy + 2 * o, 8) != 0)
          bad++;
      }
    }
  }
  // row a17: the matches manager filled from the batched run equals the replay of the direct cloud
  {
    eg3d_graph3d g;
    if (eg3d_host_replay_matches(sc, &e, &g) != 0) bad++;
    const eg3d_graph3d& h = plgmm.get_plg3d();
    if (g.n_nodes != h.n_nodes || g.n_polylines != h.n_polylines || g.n_nodes == 0 || g.n_polylines == 0 ||
        std::memcmp(g.node_X, h.node_X, 12 * g.n_nodes) != 0 || std::memcmp(g.pl_start, h.pl_start, 4 * g.n_polylines) != 0 ||
        std::memcmp(g.pl_end, h.pl_end, 4 * g.n_polylines) != 0 ||
        g.iv_off[g.n_scene_polylines] != h.iv_off[h.n_scene_polylines] || g.iv_off[g.n_scene_polylines] == 0)
      bad++;
    std::printf("replay: %llu nodes, %llu 3-D connections, %llu matched 2-D intervals\n", (unsigned long long)g.n_nodes,
                (unsigned long long)g.n_polylines, (unsigned long long)g.iv_off[g.n_scene_polylines]);
    eg3d_host_free_graph3d(&g);
  }
  // pipelines 1-2 extractor through the reference's signature: the polylines of curves 0 and 1
  {
    const uint32_t* curve = eg3d_synth_polyline_curve(syn);
    for (uint32_t cid = 0; cid < 2 && !bad; cid++) {
      std::vector<std::set<unsigned long>> compat(V);
      std::vector<uint32_t> row_off(1, 0), ids;
      for (int v = 0; v < V; v++) {
        for (uint32_t p = sc->view_pl_off[v]; p < sc->view_pl_off[v + 1]; p++)
          if (curve[p] == cid && sc->pl_valid[p] && sc->pl_vtx_off[p + 1] - sc->pl_vtx_off[p] >= 2) {
            compat[v].insert(p - sc->view_pl_off[v]);
            ids.push_back(p - sc->view_pl_off[v]);
          }
        row_off.push_back((uint32_t)ids.size());
      }
      auto via_shim = find_new_3d_points_from_compatible_polylines_expandallviews_parallel(sfm, &em, compat);
      eg3d_polyline_sets ps{1, row_off.data(), ids.empty() ? row_off.data() : ids.data()};
      eg3d_edgepoints d;
      if (eg3d_match_polyline_sets(ctx, &ps, 0, 1, 0, &d, nullptr) != EG3D_OK || via_shim.size() != d.n_points) {
        bad++;
        break;
      }
      for (uint64_t i = 0; i < d.n_points; i++)
        if (std::memcmp(&std::get<0>(via_shim[i]), d.X + 3 * i, 12) != 0 ||
            std::get<1>(via_shim[i]).size() != d.obs_off[i + 1] - d.obs_off[i])
          bad++;
      std::printf("set %u: %llu points via shim and C ABI\n", cid, (unsigned long long)d.n_points);
      eg3d_free_edgepoints(&d);
    }
  }
  // ---- the reference's EXACT call surface -------------------------------------------------
  auto same_as_direct = [&](const std::vector<new_3dpoint_plgp_matches>& got, uint64_t first, uint64_t n) {
    if (got.size() != n) return false;
    for (uint64_t k = 0; k < n; k++) {
      const uint64_t i = first + k;
      if (std::memcmp(&std::get<0>(got[k]), e.X + 3 * i, 12) != 0) return false;
      const auto& obs = std::get<1>(got[k]);
      const auto& views = std::get<2>(got[k]);
      if (obs.size() != e.obs_off[i + 1] - e.obs_off[i]) return false;
      for (size_t j = 0; j < obs.size(); j++) {
        const uint64_t o = e.obs_off[i] + (uint64_t)j;
        if (views[j] != e.obs_view[o] || obs[j].polyline_id != e.obs_pl[o] || obs[j].plp.segment_index != e.obs_seg[o] ||
            std::memcmp(&obs[j].plp.coords, e.obs_xy + 2 * o, 8) != 0)
          return false;
      }
    }
    return true;
  };
  {
    // pipelines.cpp:164 verbatim: plg_matching_from_refpoints_parallel(sfmd, em, cm, plgmm) with base-class pointers
    em.set_batching(301, 3);
    PLGPCM3ViewsPLGFollowing cm_obj(em);
    const EdgeManager* em_p = &em;
    const PLGPConsensusManager* cm_p = &cm_obj;
    PLGMatchesManager plgmm2;
    auto exact = plg_matching_from_refpoints_parallel(sfm, em_p, cm_p, plgmm2);
    auto exact_serial = plg_matching_from_refpoints(sfm, em_p, cm_p, plgmm2);
    if (!same_as_direct(exact, 0, e.n_points) || !same_as_direct(exact_serial, 0, e.n_points)) {
      std::printf("FAIL: 4-argument plg_matching_from_refpoints[_parallel] differs from the direct call\n");
      bad++;
    }
    if (plgmm2.get_plg3d().n_nodes != plgmm.get_plg3d().n_nodes || plgmm2.get_plg3d().n_polylines != plgmm.get_plg3d().n_polylines) bad++;
    // a caller-supplied consensus strategy (here: one that forwards to the path's) goes through the reference's own
    // per-point loop: plg_matching_from_refpoint(sfmd, em, cm, refpoint) for the first 40 reference points
    struct Forwarding : PLGPConsensusManager {
      PLGPCM3ViewsPLGFollowing& inner;
      int calls = 0;
      explicit Forwarding(PLGPCM3ViewsPLGFollowing& i) : inner(i) {}
      points consensus_strategy_single_point(const int a, const int b, const intersections_and_correspondences& p) override {
        return inner.consensus_strategy_single_point(a, b, p);
      }
      std::vector<points> consensus_strategy_single_point_vector(const int a, const int b,
                                                                 const intersections_and_correspondences& p) override {
        calls++;
        return inner.consensus_strategy_single_point_vector(a, b, p);
      }
    } fwd(cm_obj);
    SfMData head = sfm;
    head.numPoints_ = sfm.numPoints_ < 40 ? sfm.numPoints_ : 40;
    PLGMatchesManager plgmm3;
    auto by_point = plg_matching_from_refpoints_parallel(head, em_p, &fwd, plgmm3);
    uint64_t n_head = 0;
    while (n_head < e.n_points && e.key[4 * n_head] < (uint32_t)head.numPoints_) n_head++;
    if (!same_as_direct(by_point, 0, n_head) || fwd.calls == 0) {
      std::printf("FAIL: per-point loop with a caller-supplied consensus manager: %zu points, expected %llu\n", by_point.size(),
                  (unsigned long long)n_head);
      bad++;
    }
    std::printf("exact surface: %zu points (batched), %zu (first %d reference points through a caller's consensus manager, %d calls)\n",
                exact.size(), by_point.size(), head.numPoints_, fwd.calls);
    // the legacy segment-based virtuals throw, as the reference's PLGEdgeManager does
    bool threw = false;
    try {
      const_cast<EdgeManager*>(em_p)->detect_nearby_edge_intersections(0, 0, 10.0f);
    } catch (const NotImplementedException&) {
      threw = true;
    }
    if (!threw) bad++;
  }
  {
    // gauss_newton.hpp:20 / outliers_filtering.hpp:18-21 verbatim, on an SfMData that holds the emitted edge-points
    SfMData cloud;
    cloud.numCameras_ = V;
    cloud.imageWidth_ = sc->width;
    cloud.imageHeight_ = sc->height;
    cloud.camerasList_ = sfm.camerasList_;
    const uint64_t np = e.n_points < 20000 ? e.n_points : 20000;
    for (uint64_t i = 0; i < np; i++) {
      // every 7th point is pushed off its rays so that the filter has something to reject
      const float d = (i % 7 == 3) ? 25.0f : 0.0f;
      cloud.points_.push_back(vec3{e.X[3 * i] + d, e.X[3 * i + 1] - d, e.X[3 * i + 2]});
      std::vector<int> vs;
      std::vector<vec2> xy;
      for (uint64_t j = e.obs_off[i]; j < e.obs_off[i + 1]; j++) {
        vs.push_back(e.obs_view[j]);
        xy.push_back(vec2{e.obs_xy[2 * j], e.obs_xy[2 * j + 1]});
      }
      cloud.camViewingPointN_.push_back(vs);
      cloud.point2DoncamViewingPoint_.push_back(xy);
    }
    cloud.numPoints_ = (int)cloud.points_.size();
    // direct C-ABI answer
    std::vector<float> X(3 * np), Xo(3 * np), xy;
    std::vector<uint32_t> off(1, 0);
    std::vector<int32_t> view;
    for (uint64_t i = 0; i < np; i++) {
      X[3 * i] = cloud.points_[i].x;
      X[3 * i + 1] = cloud.points_[i].y;
      X[3 * i + 2] = cloud.points_[i].z;
      for (size_t j = 0; j < cloud.camViewingPointN_[i].size(); j++) {
        view.push_back(cloud.camViewingPointN_[i][j]);
        xy.push_back(cloud.point2DoncamViewingPoint_[i][j].x);
        xy.push_back(cloud.point2DoncamViewingPoint_[i][j].y);
      }
      off.push_back((uint32_t)view.size());
    }
    std::vector<uint8_t> inl(np);
    if (eg3d_gn_filter(ctx, X.data(), off.data(), view.data(), xy.data(), np, 2.25f, 0, Xo.data(), inl.data(), nullptr) != EG3D_OK) bad++;
    size_t n_in = 0;
    for (uint64_t i = 0; i < np; i++) n_in += inl[i];
    for (int private_ctx = 0; private_ctx < 2; private_ctx++) {
      PLGEdgeManager* saved = PLGEdgeManager::default_manager();
      if (private_ctx) PLGEdgeManager::default_manager() = nullptr;  // ./filter has no edge manager: a context from the cameras alone
      SfMData c1 = cloud;
      std::vector<bool> inliers;
      try {
        gaussNewtonFiltering(c1, inliers, 2.25f);
      } catch (const Eg3dError& ex) {
        std::printf("FAIL gaussNewtonFiltering (%s context): %s\n", private_ctx ? "private" : "registered", ex.what());
        bad++;
        PLGEdgeManager::default_manager() = saved;
        continue;
      }
      PLGEdgeManager::default_manager() = saved;
      for (uint64_t i = 0; i < np; i++) {
        if (inliers[i] != (inl[i] != 0)) bad++;
        const float* want = inl[i] ? &Xo[3 * i] : &X[3 * i];
        if (std::memcmp(&c1.points_[i], want, 12) != 0) bad++;
      }
    }
    // filter(): the four overloads against the same steps made by hand
    auto by_hand = [&](float mse, int forced) {
      std::vector<uint8_t> k(np);
      std::vector<float> Xo2(3 * np);
      if (eg3d_gn_filter(ctx, X.data(), off.data(), view.data(), xy.data(), np, mse, 0, Xo2.data(), k.data(), nullptr) != EG3D_OK) bad++;
      eg3d_host_observation_filter(V, off.data(), np, np / 2, forced, k.data());
      std::vector<vec3> pts;
      for (uint64_t i = 0; i < np; i++)
        if (k[i]) pts.push_back(vec3{Xo2[3 * i], Xo2[3 * i + 1], Xo2[3 * i + 2]});
      return pts;
    };
    auto same_pts = [&](const SfMData& a, const std::vector<vec3>& b) {
      return a.points_.size() == b.size() && a.numPoints_ == (int)b.size() && a.camViewingPointN_.size() == b.size() &&
             (b.empty() || std::memcmp(a.points_.data(), b.data(), 12 * b.size()) == 0);
    };
    const int first = (int)(np / 2);
    SfMData f1 = cloud, f2 = cloud, f3 = cloud, f4 = cloud;
    filter(f1, first);
    filter(f2, first, 1.0f);
    filter(f3, first, 5);
    filter(f4, first, 4.0f, 2);
    if (!same_pts(f1, by_hand(2.25f, -1)) || !same_pts(f2, by_hand(1.0f, -1)) || !same_pts(f3, by_hand(2.25f, 5)) ||
        !same_pts(f4, by_hand(4.0f, 2))) {
      std::printf("FAIL: filter() overloads differ from gn_filter + observation filter + removal by hand\n");
      bad++;
    }
    if (f1.points_.size() == np || f1.points_.empty()) bad++;  // the filter must have removed something, not everything
    std::printf("filter surface: %llu points, %zu Gauss-Newton inliers, filter() keeps %zu / %zu / %zu / %zu\n",
                (unsigned long long)np, n_in, f1.points_.size(), f2.points_.size(), f3.points_.size(), f4.points_.size());
  }
  // stage A through the reference's per-point entry
  auto cand = em.detect_nearby_intersections_and_correspondences_plgp(0);
  std::printf("%s points=%llu shim_batched=%zu shim_single=%zu track0_entries=%zu\n", bad ? "FAIL" : "OK",
              (unsigned long long)e.n_points, many.size(), one.size(), cand.size());
  eg3d_free_edgepoints(&e);
  eg3d_destroy(ctx);
  eg3d_synth_destroy(syn);
  return bad ? 1 : 0;
}
