// eg3d_refapi_glm.hpp — the adapters a reference maintainer needs between the reference's own glm-typed structures and
// the plain-float mirrors of eg3d_refapi.hpp. Header-only, templates over the reference's types so that
// this file needs nothing but glm: the reference's SfMData.h / types_reconstructor.hpp pull CGAL and cannot be included
// outside its build, but any struct with the same members works (duck typing):
//
//   SfMData            numPoints_, numCameras_, points_ (glm::vec3), camerasList_[v].cameraMatrix (glm::mat4, filled
//                      [row][col]), camerasPaths_, camViewingPointN_, pointsVisibleFromCamN_,
//                      point2DoncamViewingPoint_ (glm::vec2), imageWidth_, imageHeight_        SfMData.h:16-30
//   PolyLineGraph2D    polylines[p].{start, end, polyline_coords (glm::vec2)}, nodes_coords — via get_nodes_coords()
//                      or a nodes_coords member                                                polyline_graph_2d.hpp:85-119,222-294
//   cv::Mat** F        through a callable (i, j, double out[9]) -> bool (false = the 1x1 "no matrix" Mat), so that this
//                      header does not need OpenCV                                             edge_graph_3d_utilities.cpp:581-589
//   new_3dpoint_plgp_matches  tuple<glm::vec3, vector<plg_point>, vector<int>>                 polyline_graph_2d.hpp:451
//
// tests/test_refapi_glm.py compiles it against the reference tree's vendored glm (build container only).
#pragma once
#include <glm/glm.hpp>

#include "eg3d_refapi.hpp"

namespace eg3d_ref {

inline vec2 to_ref(const glm::vec2& v) { return vec2{v[0], v[1]}; }
inline vec3 to_ref(const glm::vec3& v) { return vec3{v[0], v[1], v[2]}; }
inline glm::vec2 from_ref(const vec2& v) { return glm::vec2(v.x, v.y); }
inline glm::vec3 from_ref(const vec3& v) { return glm::vec3(v.x, v.y, v.z); }

// SfMData (field-by-field copy; cameraMatrix[r][c] keeps the reference's [row][col] fill)
template <class RefSfMData>
SfMData to_ref_sfmdata(const RefSfMData& s) {
  SfMData o;
  o.numPoints_ = s.numPoints_;
  o.numCameras_ = s.numCameras_;
  o.imageWidth_ = s.imageWidth_;
  o.imageHeight_ = s.imageHeight_;
  o.points_.reserve(s.points_.size());
  for (const auto& p : s.points_) o.points_.push_back(to_ref(glm::vec3(p)));
  o.camerasList_.resize(s.camerasList_.size());
  for (size_t v = 0; v < s.camerasList_.size(); v++)
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) o.camerasList_[v].cameraMatrix[r][c] = s.camerasList_[v].cameraMatrix[r][c];
  o.camerasPaths_.assign(s.camerasPaths_.begin(), s.camerasPaths_.end());
  o.camViewingPointN_.assign(s.camViewingPointN_.begin(), s.camViewingPointN_.end());
  o.pointsVisibleFromCamN_.assign(s.pointsVisibleFromCamN_.begin(), s.pointsVisibleFromCamN_.end());
  o.point2DoncamViewingPoint_.resize(s.point2DoncamViewingPoint_.size());
  for (size_t i = 0; i < s.point2DoncamViewingPoint_.size(); i++)
    for (const auto& q : s.point2DoncamViewingPoint_[i]) o.point2DoncamViewingPoint_[i].push_back(to_ref(glm::vec2(q)));
  return o;
}
// ... and back into the reference's structure after filter() / gaussNewtonFiltering (points and tracks; the cameras of
// `dst` are left alone: the filters do not touch them)
template <class RefSfMData>
void from_ref_sfmdata(const SfMData& s, RefSfMData& dst) {
  dst.numPoints_ = s.numPoints_;
  dst.points_.clear();
  for (const vec3& p : s.points_) dst.points_.push_back(from_ref(p));
  dst.camViewingPointN_.assign(s.camViewingPointN_.begin(), s.camViewingPointN_.end());
  dst.pointsVisibleFromCamN_.assign(s.pointsVisibleFromCamN_.begin(), s.pointsVisibleFromCamN_.end());
  dst.point2DoncamViewingPoint_.clear();
  dst.point2DoncamViewingPoint_.resize(s.point2DoncamViewingPoint_.size());
  for (size_t i = 0; i < s.point2DoncamViewingPoint_.size(); i++)
    for (const vec2& q : s.point2DoncamViewingPoint_[i]) dst.point2DoncamViewingPoint_[i].push_back(from_ref(q));
}

// vector<PolyLineGraph2DHMapImpl> -> vector<eg3d_ref::PolyLineGraph2D> (polyline ids = positions, node ids kept)
template <class RefPlg>
PolyLineGraph2D to_ref_plg(const RefPlg& g) {
  PolyLineGraph2D o;
  o.polylines.resize(g.polylines.size());
  for (size_t p = 0; p < g.polylines.size(); p++) {
    o.polylines[p].start = g.polylines[p].start;
    o.polylines[p].end = g.polylines[p].end;
    for (const auto& c : g.polylines[p].polyline_coords) o.polylines[p].polyline_coords.push_back(to_ref(glm::vec2(c)));
  }
  for (const auto& n : g.nodes_coords) o.nodes_coords.push_back(to_ref(glm::vec2(n)));
  return o;
}
template <class RefPlg>
std::vector<PolyLineGraph2D> to_ref_plgs(const std::vector<RefPlg>& plgs) {
  std::vector<PolyLineGraph2D> o;
  o.reserve(plgs.size());
  for (const auto& g : plgs) o.push_back(to_ref_plg(g));
  return o;
}

// Mat** all_fundamental_matrices -> FundamentalMatrices. get(i, j, out9) copies the 3x3 CV_64F matrix row-major and
// returns true, or returns false for the 1x1 Mat the reference stores when a pair has fewer than 10 common points
// (geometric_utilities.cpp:757-781) — e.g.  [&](int i, int j, double* o) { const cv::Mat& M = F[i][j];
//   if (M.rows != 3) return false; for (int k = 0; k < 9; k++) o[k] = M.at<double>(k / 3, k % 3); return true; }
template <class GetF>
FundamentalMatrices to_ref_F(int n_views, GetF get) {
  FundamentalMatrices F((size_t)n_views, std::vector<std::array<double, 9>>((size_t)n_views));
  for (int i = 0; i < n_views; i++)
    for (int j = 0; j < n_views; j++) {
      std::array<double, 9> m{};
      if (i != j && get(i, j, m.data()))
        F[i][j] = m;
      else
        F[i][j].fill(0.0);  // all-zero = no matrix
    }
  return F;
}

// results back into the reference's tuple type: RefPlgPoint must be constructible as
// RefPlgPoint(polyline_id, segment_index, glm::vec2) (PolyLineGraph2D::plg_point, polyline_graph_2d.hpp:278-294)
template <class RefPlgPoint>
std::vector<std::tuple<glm::vec3, std::vector<RefPlgPoint>, std::vector<int>>> from_ref_points(
    const std::vector<new_3dpoint_plgp_matches>& pts) {
  std::vector<std::tuple<glm::vec3, std::vector<RefPlgPoint>, std::vector<int>>> out;
  out.reserve(pts.size());
  for (const auto& p : pts) {
    std::vector<RefPlgPoint> obs;
    for (const auto& o : std::get<1>(p)) obs.push_back(RefPlgPoint(o.polyline_id, o.plp.segment_index, from_ref(o.plp.coords)));
    out.emplace_back(from_ref(std::get<0>(p)), std::move(obs), std::get<2>(p));
  }
  return out;
}

}  // namespace eg3d_ref
