// eg3d_edge_matcher.hpp — the TOP-LEVEL seam of the reference behind its own signature:
//
//     int edge_matching(edge_matcher_input_params &emip);
//     int edge_matching(edge_matcher_input_params &emip, SfMData &sfm_data);
//
// (/root/reference/include/edgegraph3d/edge_matcher.hpp:40-42; body src/edgegraph3d/edge_matcher.cpp:60-146), with the
// reference's return convention — 0, or -1 when an input cannot be loaded — and its two written files
// (<em_out_folder>before_filtering.json, then <output_json> after filter()). Header-only C++ over the C ABI and the
// shim types of eg3d_refapi.hpp (namespace eg3d_ref), like everything a maintainer adds on the reference side
// (INTEGRATION.md). What each step of the reference's body becomes here:
//
//   parse_images(images_folder)              the photographs are only needed for their size and for the debug drawings
//                                            (out of scope): the files named by the SfM data must exist (-1 if not);
//                                            the size comes from the SfM data's intrinsics
//   parse_images(input_edges_folder)         + convert_edge_images_to_optimized_polyline_graphs (edge_matcher.cpp:84):
//                                            eg3d_plg_build_from_png per view (row N2), -1 when an edge image is
//                                            missing or unreadable
//   generate_all_fundamental_matrices        pairs of views with >= 10 common points get a matrix (the reference's rule,
//                                            exact); the matrix itself is analytic from the cameras of sfm_data_file by
//                                            default, or the build's own LMedS estimate from the tracks when
//                                            edge_matching_options().estimate_F is set (cv::findFundamentalMat(FM_LMEDS)
//                                            is randomised OpenCV code: not reproducible)
//   new PLGEdgeManager / PLGPCM3ViewsPLGFollowing / PLGMatchesManager / 4 px maps   the shim's managers (GPU context)
//   edge_reconstruction_pipeline             pipelines 1-2 start from polyline matches of the Louvain matchers (third
//                                            party, out of scope: no matches -> no points, as with an empty match
//                                            list); pipeline 3 = plg_matching_from_refpoints_parallel on the GPU; then
//                                            filter_3d_points_close_2d_array + add_3dpoints_to_sfmd
//   output_sfm_data(before_filtering.json)   eg3d_sfm_write_json with the input file passed through
//   filter(sfm_data, first_edgepoint)        eg3d_ref::filter (GPU Gauss-Newton filter + observation filter)
//   output_sfm_data(output_json)
//   draw_* (output_debug_images)             out of scope: ignored
#pragma once
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "eg3d_refapi.hpp"

namespace eg3d_ref {

// io/input/edge_graph_3d_input_params.hpp:37-48 (same fields, same order)
typedef struct edge_matcher_input_params {
  char* images_folder;
  char* input_edges_folder;
  char* em_out_folder;
  char* input_plg_folder;
  char* real_camera_poses_file;
  bool real_camera_poses_file_valid;
  char* sfm_data_file;
  unsigned long original_refpoints;
  char* output_json;
  bool output_debug_images;
} edge_matcher_input_params;

struct EdgeMatchingOptions {
  bool estimate_F = false;  // fundamental matrices estimated from the tracks (own LMedS) instead of analytic from the cameras
  bool require_images = true;  // the photographs named by the SfM data must exist in images_folder (as parse_images fails without them)
  int device = 0;
  // edge_reconstruction_pipeline (pipelines.cpp:201-246) runs three pipelines; this build runs pipeline 3 (reference points)
  // only: pipelines 1-2 consume polyline matches of the similarity-graph / Louvain matchers, which are out of scope (their
  // extractor, eg3d_match_polyline_sets, is built; examples/edge_matcher_refpoints.cpp feeds it from a file). The output can
  // therefore hold fewer edge-points than the reference's. quiet = false prints one line to stderr saying so per call;
  // after a call skipped_pipelines says which were left out.
  bool quiet = false;
  int skipped_pipelines = 0;  // bit 0: pipeline 1 (similarity graph), bit 1: pipeline 2 (closeness to refpoints) [out]
};
inline EdgeMatchingOptions& edge_matching_options() {
  static EdgeMatchingOptions o;
  return o;
}

namespace detail {
struct SfmHandle {
  eg3d_sfm* h = nullptr;
  explicit SfmHandle(eg3d_sfm* p) : h(p) {}
  ~SfmHandle() {
    if (h) eg3d_sfm_destroy(h);
  }
  SfmHandle(const SfmHandle&) = delete;
  SfmHandle& operator=(const SfmHandle&) = delete;
};
inline std::string join_path(const char* folder, const std::string& name) {
  std::string f = folder ? folder : "";
  if (!f.empty() && f.back() != '/') f += '/';
  const size_t slash = name.find_last_of('/');
  return f + (slash == std::string::npos ? name : name.substr(slash + 1));
}
inline SfMData sfmdata_from_handle(const eg3d_sfm* h) {
  SfMData s;
  const int V = eg3d_sfm_n_views(h);
  s.numCameras_ = V;
  eg3d_sfm_image_size(h, &s.imageWidth_, &s.imageHeight_);
  const float* P = eg3d_sfm_cam_P(h);
  s.camerasList_.resize(V);
  s.camerasPaths_.resize(V);
  s.pointsVisibleFromCamN_.assign(V, {});
  for (int v = 0; v < V; v++) {
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) s.camerasList_[v].cameraMatrix[r][c] = P[(size_t)v * 16 + r * 4 + c];
    const char* p = eg3d_sfm_image_path(h, v);
    s.camerasPaths_[v] = p ? p : "";
  }
  eg3d_seeds sd;
  eg3d_sfm_seeds(h, &sd);
  const float* X = eg3d_sfm_points(h);
  s.numPoints_ = (int)sd.n_seeds;
  s.points_.resize(sd.n_seeds);
  s.camViewingPointN_.resize(sd.n_seeds);
  s.point2DoncamViewingPoint_.resize(sd.n_seeds);
  for (uint32_t i = 0; i < sd.n_seeds; i++) {
    s.points_[i] = vec3{X[3 * i], X[3 * i + 1], X[3 * i + 2]};
    for (uint32_t k = sd.trk_off[i]; k < sd.trk_off[i + 1]; k++) {
      s.camViewingPointN_[i].push_back(sd.trk_view[k]);
      s.point2DoncamViewingPoint_[i].push_back(vec2{sd.trk_xy[2 * k], sd.trk_xy[2 * k + 1]});
      s.pointsVisibleFromCamN_[sd.trk_view[k]].push_back((int)i);
    }
  }
  return s;
}
}  // namespace detail

// read_sfm_data (io/input/input_reader: OpenMvgParser::parse, OpenMvgParser.cpp:39-301); throws Eg3dError when the file
// cannot be parsed
inline SfMData read_sfm_data(const char* sfm_data_file) {
  detail::SfmHandle h(eg3d_sfm_read_json(sfm_data_file));
  if (!h.h) throw Eg3dError(EG3D_ERR_ARG, std::string("read_sfm_data: ") + (sfm_data_file ? sfm_data_file : "(null)"));
  return detail::sfmdata_from_handle(h.h);
}

// output_sfm_data(original_sfm_data_file, sfmd, output_file) (output_sfm_data.cpp:186-229): the input document with its
// "structure" replaced by the points of sfmd. Returns 0, or non-zero when a file cannot be read / written.
inline int output_sfm_data(const char* original_sfm_data_file, const SfMData& sfmd, const std::string& output_file) {
  detail::SfmHandle h(eg3d_sfm_create(sfmd.numCameras_, sfmd.imageWidth_, sfmd.imageHeight_));
  if (!h.h) return -1;
  std::vector<int32_t> views;
  std::vector<float> xy;
  for (size_t i = 0; i < sfmd.points_.size(); i++) {
    views.assign(sfmd.camViewingPointN_[i].begin(), sfmd.camViewingPointN_[i].end());
    xy.clear();
    for (const vec2& p : sfmd.point2DoncamViewingPoint_[i]) {
      xy.push_back(p.x);
      xy.push_back(p.y);
    }
    const float X[3] = {sfmd.points_[i].x, sfmd.points_[i].y, sfmd.points_[i].z};
    if (eg3d_sfm_add_point(h.h, X, (int)views.size(), views.data(), xy.data()) != 0) return -1;
  }
  return eg3d_sfm_write_json(h.h, original_sfm_data_file, output_file.c_str());
}

// convert_edge_images_to_optimized_polyline_graphs over the edge images named by the SfM data
// (io/input/convert_edge_images_pixel_to_segment.cpp:868-892 -> :294-626). false when an image cannot be read or
// the images differ in size.
inline bool convert_edge_images_to_optimized_polyline_graphs(const char* input_edges_folder, const SfMData& sfmd,
                                                             std::vector<PolyLineGraph2D>& plgs, int& width, int& height) {
  plgs.clear();
  plgs.resize((size_t)sfmd.numCameras_);
  width = height = 0;
  // the views are independent: all of them are built at once on the host's cores (eg3d_plg_build_views_from_png; one after
  // the other — the reference's loop — 25 dtu006-sized edge maps take 3 s, sixty times the GPU's share of the whole call)
  std::vector<std::string> paths;
  std::vector<const char*> cpaths;
  for (int v = 0; v < sfmd.numCameras_; v++) paths.push_back(detail::join_path(input_edges_folder, sfmd.camerasPaths_[v]));
  for (const std::string& p : paths) cpaths.push_back(p.c_str());
  std::vector<eg3d_plg_view> views((size_t)sfmd.numCameras_);
  if (eg3d_plg_build_views_from_png(cpaths.data(), sfmd.numCameras_, &width, &height, views.data()) != 0) return false;
  for (int v = 0; v < sfmd.numCameras_; v++) {
    eg3d_plg_view& g = views[(size_t)v];
    PolyLineGraph2D& out = plgs[(size_t)v];
    out.polylines.resize(g.n_polylines);
    for (uint32_t p = 0; p < g.n_polylines; p++) {
      out.polylines[p].start = g.pl_start[p];
      out.polylines[p].end = g.pl_end[p];
      for (uint32_t k = g.pl_vtx_off[p]; k < g.pl_vtx_off[p + 1]; k++)
        out.polylines[p].polyline_coords.push_back(vec2{g.vtx_xy[2 * k], g.vtx_xy[2 * k + 1]});
    }
    out.nodes_coords.resize(g.n_nodes);
    for (uint32_t n = 0; n < g.n_nodes; n++) out.nodes_coords[n] = vec2{g.node_xy[2 * n], g.node_xy[2 * n + 1]};
    eg3d_plg_view_free(&g);
  }
  return true;
}

// filter_3d_points_close_2d_array (filtering_close_plgps.cpp:99-124): keeps the points no earlier kept point is within
// 3 px of in every common view; order preserved
inline std::vector<new_3dpoint_plgp_matches> filter_3d_points_close_2d_array(int n_views, int width, int height,
                                                                             const std::vector<new_3dpoint_plgp_matches>& p3ds) {
  std::vector<float> X, xy;
  std::vector<uint64_t> off(1, 0);
  std::vector<int32_t> view;
  std::vector<uint32_t> pl, seg, key;
  for (const auto& p : p3ds) {
    const vec3& x = std::get<0>(p);
    X.insert(X.end(), {x.x, x.y, x.z});
    key.insert(key.end(), {0u, 0u, 0u, 0u});
    const auto& obs = std::get<1>(p);
    const auto& vs = std::get<2>(p);
    for (size_t j = 0; j < obs.size(); j++) {
      view.push_back(vs[j]);
      pl.push_back((uint32_t)obs[j].polyline_id);
      seg.push_back((uint32_t)obs[j].plp.segment_index);
      xy.push_back(obs[j].plp.coords.x);
      xy.push_back(obs[j].plp.coords.y);
    }
    off.push_back((uint64_t)view.size());
  }
  if (X.empty()) X.assign(3, 0.f);
  if (key.empty()) key.assign(4, 0);
  if (view.empty()) {
    view.assign(1, 0);
    pl.assign(1, 0);
    seg.assign(1, 0);
    xy.assign(2, 0.f);
  }
  eg3d_edgepoints e;
  std::memset(&e, 0, sizeof(e));
  e.n_points = p3ds.size();
  e.n_obs = off.back();
  e.X = X.data();
  e.obs_off = off.data();
  e.obs_view = view.data();
  e.obs_pl = pl.data();
  e.obs_seg = seg.data();
  e.obs_xy = xy.data();
  e.key = key.data();
  std::vector<uint8_t> keep(p3ds.size() ? p3ds.size() : 1);
  if (eg3d_host_filter_close_2d(n_views, width, height, &e, keep.data()) != 0)
    throw Eg3dError(EG3D_ERR_ARG, "filter_3d_points_close_2d_array");
  std::vector<new_3dpoint_plgp_matches> out;
  for (size_t i = 0; i < p3ds.size(); i++)
    if (keep[i]) out.push_back(p3ds[i]);
  return out;
}

// add_3dpoints_to_sfmd (output_utilities.cpp:96-111)
inline void add_3dpoints_to_sfmd(SfMData& sfmd, const std::vector<new_3dpoint_plgp_matches>& p3ds) {
  for (const auto& p3d : p3ds) {
    const int new_point_id = (int)sfmd.points_.size();
    sfmd.points_.push_back(std::get<0>(p3d));
    for (const int cam_id : std::get<2>(p3d)) sfmd.pointsVisibleFromCamN_[(size_t)cam_id].push_back(new_point_id);
    std::vector<vec2> coords;
    for (const auto& o : std::get<1>(p3d)) coords.push_back(o.plp.coords);
    sfmd.point2DoncamViewingPoint_.push_back(coords);
    sfmd.camViewingPointN_.push_back(std::get<2>(p3d));
  }
  sfmd.numPoints_ = (int)sfmd.points_.size();
}

// generate_all_fundamental_matrices (geometric_utilities.cpp:754-820) — see the header comment for what is exact
inline bool generate_all_fundamental_matrices(const char* sfm_data_file, const SfMData& sfmd, FundamentalMatrices& F) {
  const int V = sfmd.numCameras_;
  std::vector<uint32_t> off(1, 0);
  std::vector<int32_t> view;
  std::vector<float> xy;
  for (int i = 0; i < sfmd.numPoints_; i++) {
    for (size_t k = 0; k < sfmd.camViewingPointN_[(size_t)i].size(); k++) {
      view.push_back(sfmd.camViewingPointN_[(size_t)i][k]);
      xy.push_back(sfmd.point2DoncamViewingPoint_[(size_t)i][k].x);
      xy.push_back(sfmd.point2DoncamViewingPoint_[(size_t)i][k].y);
    }
    off.push_back((uint32_t)view.size());
  }
  if (view.empty()) {
    view.push_back(0);
    xy.assign(2, 0.f);
  }
  std::vector<double> Fm((size_t)V * V * 9, 0.0);
  std::vector<uint8_t> valid((size_t)V * V, 0);
  if (edge_matching_options().estimate_F) {
    if (eg3d_host_estimate_F(V, (uint64_t)sfmd.numPoints_, off.data(), view.data(), xy.data(), 1, 0xE63D2018ull, Fm.data(), valid.data(),
                             nullptr) < 0)
      return false;
  } else {
    detail::SfmHandle h(eg3d_sfm_read_json(sfm_data_file));  // the cameras (focal, rotation, centre) live in the file
    if (!h.h || eg3d_sfm_n_views(h.h) != V || eg3d_sfm_analytic_F(h.h, Fm.data(), valid.data()) != 0) return false;
    std::vector<uint8_t> rule((size_t)V * V, 0);
    if (eg3d_host_estimate_F(V, (uint64_t)sfmd.numPoints_, off.data(), view.data(), xy.data(), 0, 0, nullptr, rule.data(), nullptr) < 0)
      return false;
    for (size_t k = 0; k < valid.size(); k++) valid[k] &= rule[k];
  }
  F.assign((size_t)V, std::vector<std::array<double, 9>>((size_t)V));
  for (int i = 0; i < V; i++)
    for (int j = 0; j < V; j++)
      for (int k = 0; k < 9; k++) F[(size_t)i][(size_t)j][(size_t)k] = valid[(size_t)i * V + j] ? Fm[((size_t)i * V + j) * 9 + k] : 0.0;
  return true;
}

// edge_matcher.cpp:66-146
inline int edge_matching(edge_matcher_input_params& emip, SfMData& sfm_data) {
  if (edge_matching_options().require_images && emip.images_folder)
    for (const std::string& name : sfm_data.camerasPaths_) {
      FILE* f = std::fopen(detail::join_path(emip.images_folder, name).c_str(), "rb");
      if (!f) return -1;  // something went wrong in loading
      std::fclose(f);
    }
  std::vector<PolyLineGraph2D> plgs;
  int w = 0, h = 0;
  if (!convert_edge_images_to_optimized_polyline_graphs(emip.input_edges_folder, sfm_data, plgs, w, h)) return -1;
  if (sfm_data.imageWidth_ <= 0 || sfm_data.imageHeight_ <= 0) {
    sfm_data.imageWidth_ = w;
    sfm_data.imageHeight_ = h;
  }
  FundamentalMatrices F;
  if (!generate_all_fundamental_matrices(emip.sfm_data_file, sfm_data, F)) return -1;

  PLGMatchesManager plgmm;
  std::unique_ptr<PLGEdgeManager> em(new PLGEdgeManager(sfm_data, F, plgs, 10.0f, 3.0f, edge_matching_options().device));
  if (em->last_status() != EG3D_OK) throw Eg3dError(em->last_status(), std::string("edge_matching: ") + eg3d_last_error());
  std::unique_ptr<PLGPCM3ViewsPLGFollowing> cm(new PLGPCM3ViewsPLGFollowing(*em));
  const int first_edgepoint = (int)sfm_data.points_.size();

  // edge_reconstruction_pipeline (pipelines.cpp:201-246): pipelines 1-2 have no polyline matches here (their matchers are
  // out of scope); pipeline 3
  edge_matching_options().skipped_pipelines = 3;
  if (!edge_matching_options().quiet)
    std::fprintf(stderr, "eg3d edge_matching: pipelines 1-2 (polyline matches of the similarity-graph / Louvain matchers) are not "
                         "part of this build; running pipeline 3 (reference points) only: the output may hold fewer edge-points "
                         "than the reference's (INTEGRATION.md)\n");
  std::vector<new_3dpoint_plgp_matches> p3ds = plg_matching_from_refpoints_parallel(sfm_data, em.get(), cm.get(), plgmm);
  const std::vector<new_3dpoint_plgp_matches> filtered_p3ds =
      filter_3d_points_close_2d_array(sfm_data.numCameras_, sfm_data.imageWidth_, sfm_data.imageHeight_, p3ds);
  add_3dpoints_to_sfmd(sfm_data, filtered_p3ds);

  const std::string out_folder = emip.em_out_folder ? emip.em_out_folder : "";
  if (output_sfm_data(emip.sfm_data_file, sfm_data, out_folder + "before_filtering.json") != 0)
    throw Eg3dError(EG3D_ERR_ARG, "edge_matching: cannot write before_filtering.json");
  filter(sfm_data, first_edgepoint);
  if (output_sfm_data(emip.sfm_data_file, sfm_data, emip.output_json ? emip.output_json : "") != 0)
    throw Eg3dError(EG3D_ERR_ARG, "edge_matching: cannot write the output file");
  return 0;
}

// edge_matcher.cpp:60-64
inline int edge_matching(edge_matcher_input_params& emip) {
  SfMData sfmd = read_sfm_data(emip.sfm_data_file);
  return edge_matching(emip, sfmd);
}

}  // namespace eg3d_ref
