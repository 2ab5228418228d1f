// examples/edge_matching_main.cpp — the reference's top-level call, edge_matching(emip) (edge_matcher.cpp:60-146), through
// the shim of include/eg3d_edge_matcher.hpp: edge images + OpenMVG JSON in, OpenMVG JSON with the edge-points out.
//   edge_matching_main <images folder> <edge images folder> <input.json> <out folder/> <output.json> [--estimate-F]
// (the longer examples/edge_matcher_refpoints.cpp drives the same steps one by one over the C ABI, with the polyline
// graphs in a container file and the polyline matches of pipelines 1-2 from files)
#include <cstdio>
#include <cstring>

#include "eg3d_edge_matcher.hpp"

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s <images folder> <edge images folder> <input.json> <out folder/> <output.json> [--estimate-F]\n", argv[0]);
    return 2;
  }
  eg3d_ref::edge_matcher_input_params emip;
  std::memset(&emip, 0, sizeof(emip));
  emip.images_folder = argv[1];
  emip.input_edges_folder = argv[2];
  emip.sfm_data_file = argv[3];
  emip.em_out_folder = argv[4];
  emip.output_json = argv[5];
  eg3d_ref::edge_matching_options().estimate_F = argc > 6 && std::strcmp(argv[6], "--estimate-F") == 0;
  try {
    const int rc = eg3d_ref::edge_matching(emip);
    std::printf("edge_matching returned %d\n", rc);
    return rc == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "edge_matching failed: %s\n", e.what());
    return 1;
  }
}
