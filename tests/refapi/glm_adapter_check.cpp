// Build-container-only check of include/eg3d_refapi_glm.hpp: structures shaped like the reference's (same member
// names, glm types from the reference tree's vendored glm — SfMData.h:16-30, polyline_graph_2d.hpp:85-119,278-294) go
// through the adapters and back unchanged. No GPU needed (nothing is matched). Compile: g++ -std=c++17 -I include
// -I /root/reference/external/glm tests/refapi/glm_adapter_check.cpp
#include <cstdio>
#include <cstring>
#include <string>

#include "eg3d_refapi_glm.hpp"

namespace likeref {  // members as in the reference's headers
struct CameraType {
  glm::mat3 intrinsics, rotation;
  glm::vec3 translation;
  glm::mat4 cameraMatrix;
  glm::vec3 center;
};
struct SfMData {
  int numPoints_, numCameras_;
  std::vector<glm::vec3> points_;
  std::vector<CameraType> camerasList_;
  std::vector<std::string> camerasPaths_;
  std::vector<std::vector<int>> camViewingPointN_;
  std::vector<std::vector<int>> pointsVisibleFromCamN_;
  std::vector<std::vector<glm::vec2>> point2DoncamViewingPoint_;
  int imageWidth_, imageHeight_;
};
struct PolyLineGraph2D {
  struct polyline {
    unsigned long start, end;
    std::vector<glm::vec2> polyline_coords;
  };
  struct plg_point {
    unsigned long polyline_id;
    unsigned long segment_index;
    glm::vec2 coords;
    plg_point(unsigned long p, unsigned long s, const glm::vec2& c) : polyline_id(p), segment_index(s), coords(c) {}
  };
  std::vector<polyline> polylines;
  std::vector<glm::vec2> nodes_coords;
};
}  // namespace likeref

int main() {
  likeref::SfMData s;
  s.numPoints_ = 2;
  s.numCameras_ = 2;
  s.imageWidth_ = 1600;
  s.imageHeight_ = 1200;
  s.points_ = {glm::vec3(1.5f, -2.25f, 3.f), glm::vec3(0.1f, 0.2f, 0.3f)};
  s.camerasList_.resize(2);
  for (int v = 0; v < 2; v++)
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) s.camerasList_[v].cameraMatrix[r][c] = 100.f * v + 4.f * r + c;  // [row][col] fill (Q6)
  s.camerasPaths_ = {"0000.png", "0001.png"};
  s.camViewingPointN_ = {{0, 1}, {1}};
  s.pointsVisibleFromCamN_ = {{0}, {0, 1}};
  s.point2DoncamViewingPoint_ = {{glm::vec2(10.f, 20.f), glm::vec2(30.5f, 40.25f)}, {glm::vec2(1.f, 2.f)}};
  eg3d_ref::SfMData r = eg3d_ref::to_ref_sfmdata(s);
  int bad = 0;
  bad += r.numPoints_ != 2 || r.points_[0].y != -2.25f || r.camerasList_[1].cameraMatrix[2][3] != 111.f ||
         r.camerasList_[0].cameraMatrix[3][0] != 12.f || r.point2DoncamViewingPoint_[0][1].y != 40.25f ||
         r.camViewingPointN_[0][1] != 1 || r.camerasPaths_[1] != "0001.png" || r.imageWidth_ != 1600;
  r.points_.pop_back();  // what filter() does: the structure shrinks
  r.camViewingPointN_.pop_back();
  r.point2DoncamViewingPoint_.pop_back();
  r.numPoints_ = 1;
  eg3d_ref::from_ref_sfmdata(r, s);
  bad += s.numPoints_ != 1 || s.points_.size() != 1 || s.points_[0][2] != 3.f || s.point2DoncamViewingPoint_.size() != 1 ||
         s.point2DoncamViewingPoint_[0][1][0] != 30.5f || s.camerasList_.size() != 2;

  likeref::PolyLineGraph2D g;
  g.polylines.push_back({3, 7, {glm::vec2(1.f, 1.f), glm::vec2(2.f, 3.f), glm::vec2(5.f, 8.f)}});
  g.nodes_coords.assign(8, glm::vec2(-1.f, -1.f));
  g.nodes_coords[3] = glm::vec2(1.f, 1.f);
  g.nodes_coords[7] = glm::vec2(5.f, 8.f);
  std::vector<likeref::PolyLineGraph2D> gs(2, g);
  auto rg = eg3d_ref::to_ref_plgs(gs);
  bad += rg.size() != 2 || rg[1].polylines[0].end != 7 || rg[0].polylines[0].polyline_coords[2].y != 8.f || !rg[0].is_valid_polyline(0);

  auto F = eg3d_ref::to_ref_F(2, [](int i, int j, double* o) {
    if (i == 1) return false;  // "1x1 Mat": no matrix for pairs starting at view 1
    for (int k = 0; k < 9; k++) o[k] = 10.0 * i + j + 0.125 * k;
    return true;
  });
  bad += F[0][1][8] != 2.0 || F[1][0][0] != 0.0 || F[0][0][4] != 0.0;

  std::vector<eg3d_ref::new_3dpoint_plgp_matches> pts;
  pts.emplace_back(eg3d_ref::vec3{1.f, 2.f, 3.f}, std::vector<eg3d_ref::PolyLineGraph2D::plg_point>{{5, {2, {7.5f, 8.5f}}}},
                   std::vector<int>{4});
  auto back = eg3d_ref::from_ref_points<likeref::PolyLineGraph2D::plg_point>(pts);
  bad += back.size() != 1 || std::get<0>(back[0])[1] != 2.f || std::get<1>(back[0])[0].polyline_id != 5 ||
         std::get<1>(back[0])[0].segment_index != 2 || std::get<1>(back[0])[0].coords[1] != 8.5f || std::get<2>(back[0])[0] != 4;
  std::printf(bad ? "GLM-ADAPTER-FAIL %d\n" : "GLM-ADAPTER-OK\n", bad);
  return bad ? 1 : 0;
}
