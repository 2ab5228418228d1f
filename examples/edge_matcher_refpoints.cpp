// examples/edge_matcher_refpoints.cpp — the reference's `edge_matcher` run, all three stages of
// edge_reconstruction_pipeline (pipelines.cpp:201-246), end to end in C++ over the C ABI
// (what src/edgegraph3d/edge_matcher.cpp:57-182 + pipelines.cpp do):
//
//   OpenMVG JSON (cameras + SfM points)  +  polyline graphs of every view
//     -> fundamental matrices            (edge_matcher.cpp:96 -> geometric_utilities.cpp:754-820): pairs of views
//                                        with >= 10 common SfM points get a matrix, the others none (exact rule);
//                                        the matrix is analytic from the cameras by default, or the build's own
//                                        least-median-of-squares estimate from the tracks with --estimate-F
//                                        (the reference's cv::findFundamentalMat(FM_LMEDS) is randomised OpenCV
//                                        code: not reproducible)
//     -> [pipeline 1] polyline matches of the similarity-graph matcher  (pipelines.cpp:219 -> :68-111)
//     -> [pipeline 2] polyline matches by closeness to reference points (pipelines.cpp:223 -> :113-158)
//                                        each match = one set of potentially compatible polylines, given in a
//                                        file (--sets1 / --sets2; the matchers themselves — Louvain clustering,
//                                        third party — are out of scope): find_new_3d_points_from_compatible_
//                                        polylines_expandallviews_parallel per match   GPU: eg3d_match_polyline_sets
//     -> [pipeline 3] plg_matching_from_refpoints (pipelines.cpp:227 -> :164)    GPU: eg3d_match_refpoints
//                                        the three results are concatenated in that order (pipelines.cpp:219-227;
//                                        the matches manager stays empty until pipeline 3 in the parallel build, so
//                                        the stages do not see each other's matches)
//     -> filter_3d_points_close_2d_array (pipelines.cpp:236)            host, over the concatenation
//     -> add_3dpoints_to_sfmd            (edge_matcher.cpp:158)         host
//     -> [./filter -e] gaussNewtonFiltering + observation filter + removeOutliers   GPU + host
//     -> output_sfm_data                 (edge_matcher.cpp:169)         OpenMVG JSON out
//
// Usage:
//   edge_matcher_refpoints --make-synthetic <config 0..4> <dir>      writes <dir>/input.json, <dir>/plgs.bin
//   edge_matcher_refpoints --make-plgs <plgs.bin> <edge image 0.png> <edge image 1.png> ...
//        builds the polyline graph of every binary edge image (view i = i-th image; SURVEY N2,
//        edge_matcher.cpp:84-94 convert_edge_images_to_optimized_polyline_graphs) and writes the container
//   edge_matcher_refpoints <dir>/input.json <dir>/plgs.bin <out.json> [--filter] [--estimate-F] [--all-pairs]
//                          [--sets1 <polyline matches of pipeline 1>] [--sets2 <... of pipeline 2>]
//        a match file is text: "eg3d-polyline-sets 1", then "<n_sets> <n_views>", then one line per (set, view):
//        "<count> <polyline id> ..." with view-local ids (the reference's vector<set<ulong>> per match)
//        (--make-synthetic also writes <dir>/sets1.txt and <dir>/sets2.txt: the polylines of 3-D curves 0-2 / 3-5)
//
// The polyline graphs travel in a container file so that the expensive one-off construction from the
// edge images (--make-plgs) is separate from the matching run. Build (see tests/test_gpu_edge_cases.py):
//   g++ -std=c++17 -I include examples/edge_matcher_refpoints.cpp -Ledgegraph3d_amd -leg3d -leg3d_host ...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "eg3d.h"
#include "eg3d_host.h"

static int fail(const char* what) {
  std::fprintf(stderr, "edge_matcher_refpoints: %s (%s)\n", what, eg3d_last_error());
  return 1;
}

// ---- polyline match files (input of pipelines 1-2)
struct MatchSets {
  uint32_t n_sets = 0;
  std::vector<uint32_t> row_off{0}, ids;
};
static bool read_match_sets(const char* path, int n_views, MatchSets& m) {
  FILE* f = std::fopen(path, "r");
  if (!f) return false;
  char tag[64];
  int ver = 0;
  unsigned ns = 0, nv = 0;
  bool ok = std::fscanf(f, "%63s %d %u %u", tag, &ver, &ns, &nv) == 4 && std::strcmp(tag, "eg3d-polyline-sets") == 0 && ver == 1 &&
            (int)nv == n_views && ns < (1u << 24);
  for (uint64_t r = 0; ok && r < (uint64_t)ns * nv; r++) {
    unsigned k = 0;
    ok = std::fscanf(f, "%u", &k) == 1 && k < (1u << 24);
    unsigned long prev = 0;
    for (unsigned i = 0; ok && i < k; i++) {
      unsigned long id = 0;
      ok = std::fscanf(f, "%lu", &id) == 1 && id <= 0xfffffffful && (i == 0 || id > prev);  // a set: ascending, no repeats
      prev = id;
      m.ids.push_back((uint32_t)id);
    }
    m.row_off.push_back((uint32_t)m.ids.size());
  }
  std::fclose(f);
  m.n_sets = ns;
  if (m.ids.empty()) m.ids.push_back(0);
  return ok;
}
static bool write_match_sets(const std::string& path, const eg3d_scene* sc, const uint32_t* curve, uint32_t c0, uint32_t c1) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  std::fprintf(f, "eg3d-polyline-sets 1\n%u %d\n", c1 - c0, sc->n_views);
  for (uint32_t c = c0; c < c1; c++)
    for (int v = 0; v < sc->n_views; v++) {
      std::vector<uint32_t> ids;
      for (uint32_t p = sc->view_pl_off[v]; p < sc->view_pl_off[v + 1]; p++)
        if (curve[p] == c && sc->pl_valid[p] && sc->pl_vtx_off[p + 1] - sc->pl_vtx_off[p] >= 2) ids.push_back(p - sc->view_pl_off[v]);
      std::fprintf(f, "%zu", ids.size());
      for (uint32_t id : ids) std::fprintf(f, " %u", id);
      std::fprintf(f, "\n");
    }
  return std::fclose(f) == 0;
}

// ---- the clouds of the three stages, concatenated in stage order (host arrays in the eg3d_edgepoints layout)
struct Cloud {
  std::vector<float> X, xy;
  std::vector<uint64_t> off{0};
  std::vector<int32_t> view;
  std::vector<uint32_t> pl, seg, key;
  void append(const eg3d_edgepoints& e) {
    const uint64_t base = (uint64_t)view.size();
    X.insert(X.end(), e.X, e.X + 3 * e.n_points);
    key.insert(key.end(), e.key, e.key + 4 * e.n_points);
    for (uint64_t i = 0; i < e.n_points; i++) off.push_back(base + e.obs_off[i + 1]);
    view.insert(view.end(), e.obs_view, e.obs_view + e.n_obs);
    pl.insert(pl.end(), e.obs_pl, e.obs_pl + e.n_obs);
    seg.insert(seg.end(), e.obs_seg, e.obs_seg + e.n_obs);
    xy.insert(xy.end(), e.obs_xy, e.obs_xy + 2 * e.n_obs);
  }
  eg3d_edgepoints view_as_edgepoints() {
    eg3d_edgepoints e;
    std::memset(&e, 0, sizeof(e));
    e.n_points = off.size() - 1;
    e.n_obs = view.size();
    if (X.empty()) X.assign(3, 0.f);  // never hand out null arrays
    if (key.empty()) key.assign(4, 0);
    if (view.empty()) {
      view.assign(1, 0);
      pl.assign(1, 0);
      seg.assign(1, 0);
      xy.assign(2, 0.f);
    }
    e.X = X.data();
    e.obs_off = off.data();
    e.obs_view = view.data();
    e.obs_pl = pl.data();
    e.obs_seg = seg.data();
    e.obs_xy = xy.data();
    e.key = key.data();
    return e;
  }
};

static int make_synthetic(int cfg_index, const std::string& dir) {
  eg3d_synth_config cfg;
  eg3d_synth_default_config(&cfg, cfg_index);
  eg3d_synth* syn = eg3d_synth_create(&cfg);
  if (!syn) return fail("synthetic scene");
  const eg3d_scene* sc = eg3d_synth_scene(syn);
  const eg3d_seeds* sd = eg3d_synth_seeds(syn);
  const float* truth = eg3d_synth_seed_truth(syn);
  eg3d_sfm* sfm = eg3d_sfm_create(sc->n_views, sc->width, sc->height);
  for (int v = 0; v < sc->n_views; v++) {
    float f, px, py, R[9], C[3];
    eg3d_synth_camera(syn, v, &f, &px, &py, R, C);
    char name[64];
    std::snprintf(name, sizeof(name), "%04d.png", v);
    eg3d_sfm_set_camera(sfm, v, f, px, py, R, C, name);
  }
  for (uint32_t i = 0; i < sd->n_seeds; i++) {
    const uint32_t a = sd->trk_off[i], b = sd->trk_off[i + 1];
    eg3d_sfm_add_point(sfm, truth + 3 * i, (int)(b - a), sd->trk_view + a, sd->trk_xy + 2 * a);
  }
  int rc = eg3d_sfm_write_json(sfm, nullptr, (dir + "/input.json").c_str());
  if (rc == 0) rc = eg3d_plg_write((dir + "/plgs.bin").c_str(), sc);
  // stand-ins for the polyline matchers' output: the polylines generated from 3-D curves 0-2 (pipeline 1) and 3-5 (pipeline 2)
  const uint32_t nc = eg3d_synth_n_curves(syn);
  const uint32_t c1 = nc < 3 ? nc : 3, c2 = nc < 6 ? nc : 6;
  if (rc == 0 && !(write_match_sets(dir + "/sets1.txt", sc, eg3d_synth_polyline_curve(syn), 0, c1) &&
                   write_match_sets(dir + "/sets2.txt", sc, eg3d_synth_polyline_curve(syn), c1, c2)))
    rc = -1;
  std::printf("wrote %s/input.json (%d views, %u points) and %s/plgs.bin (%u polylines)\n", dir.c_str(), sc->n_views,
              sd->n_seeds, dir.c_str(), sc->view_pl_off[sc->n_views]);
  eg3d_sfm_destroy(sfm);
  eg3d_synth_destroy(syn);
  return rc == 0 ? 0 : fail("writing the synthetic inputs");
}

static int make_plgs(const char* out_path, int n, char** images) {
  std::vector<eg3d_plg_view> views((size_t)n);
  int W = 0, H = 0;
  // all views at once, on the host's cores (the views are independent)
  const int bad = eg3d_plg_build_views_from_png(images, n, &W, &H, views.data());
  if (bad != 0) {
    std::fprintf(stderr, "edge_matcher_refpoints: cannot build the polyline graph of %s (unreadable, or not the size of the first image)\n",
                 bad < 0 && -bad - 1 < n ? images[-bad - 1] : "?");
    return 1;
  }
  eg3d_plg* g = eg3d_plg_from_views(n, W, H, views.data());
  uint64_t n_pl = 0;
  for (auto& v : views) {
    n_pl += v.n_polylines;
    eg3d_plg_view_free(&v);
  }
  if (!g || eg3d_plg_write(out_path, eg3d_plg_scene(g)) != 0) return fail("writing the polyline graphs");
  std::printf("wrote %s: %d views of %dx%d, %llu polylines\n", out_path, n, W, H, (unsigned long long)n_pl);
  eg3d_plg_destroy(g);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 4 && std::strcmp(argv[1], "--make-synthetic") == 0) return make_synthetic(std::atoi(argv[2]), argv[3]);
  if (argc >= 4 && std::strcmp(argv[1], "--make-plgs") == 0) return make_plgs(argv[2], argc - 3, argv + 3);
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <input.json> <plgs.bin> <out.json> [--filter] [--estimate-F] [--all-pairs]\n", argv[0]);
    return 2;
  }
  bool do_filter = false, estimate_F = false, all_pairs = false;
  const char* sets_path[2] = {nullptr, nullptr};
  for (int a = 4; a < argc; a++) {
    do_filter |= std::strcmp(argv[a], "--filter") == 0;
    estimate_F |= std::strcmp(argv[a], "--estimate-F") == 0;
    all_pairs |= std::strcmp(argv[a], "--all-pairs") == 0;  // analytic F for every pair, ignoring the 10-point rule
    if (std::strcmp(argv[a], "--sets1") == 0 && a + 1 < argc) sets_path[0] = argv[++a];
    else if (std::strcmp(argv[a], "--sets2") == 0 && a + 1 < argc) sets_path[1] = argv[++a];
  }

  // wall time of every stage of the run, host stages included (--times prints them; stderr)
  bool print_times = false;
  for (int a = 4; a < argc; a++) print_times |= std::strcmp(argv[a], "--times") == 0;
  auto t_last = std::chrono::steady_clock::now();
  const auto t_begin = t_last;
  auto lap = [&](const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (print_times)
      std::fprintf(stderr, "[time] %-44s %9.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // ---- inputs (edge_matcher.cpp:64-95)
  eg3d_sfm* sfm = eg3d_sfm_read_json(argv[1]);
  lap("read the OpenMVG JSON");
  if (!sfm) return fail("reading the OpenMVG file");
  eg3d_plg* plg = eg3d_plg_read(argv[2]);
  if (!plg) return fail("reading the polyline graphs");
  lap("read the polyline graphs");
  const int V = eg3d_sfm_n_views(sfm);
  eg3d_scene sc = *eg3d_plg_scene(plg);
  if (sc.n_views != V) return fail("views of the SfM file and of the polyline graphs differ");
  std::vector<double> F((size_t)V * V * 9);
  std::vector<uint8_t> Fv((size_t)V * V);
  if (estimate_F) {
    const int bad = eg3d_sfm_estimate_F(sfm, 1, 0xE63D2018ull, F.data(), Fv.data(), nullptr);
    if (bad < 0) return fail("fundamental matrices");
    if (bad > 0) std::printf("%d view pairs: estimate failed, left without a matrix\n", bad);
  } else {
    if (eg3d_sfm_analytic_F(sfm, F.data(), Fv.data()) != 0) return fail("fundamental matrices");
    if (!all_pairs) {  // the reference's rule for which pairs have a matrix at all
      std::vector<uint8_t> rule((size_t)V * V);
      if (eg3d_sfm_estimate_F(sfm, 0, 0, nullptr, rule.data(), nullptr) < 0) return fail("fundamental matrices");
      for (size_t k = 0; k < rule.size(); k++) Fv[k] &= rule[k];
    }
  }
  size_t n_pairs = 0;
  for (uint8_t v : Fv) n_pairs += v;
  std::printf("fundamental matrices for %zu of %d ordered view pairs (%s)\n", n_pairs, V * (V - 1),
              estimate_F ? "least-median-of-squares estimate from the tracks" : "analytic from the cameras");
  sc.cam_P = eg3d_sfm_cam_P(sfm);
  sc.F = F.data();
  sc.F_valid = Fv.data();
  const uint64_t first_edgepoint = eg3d_sfm_n_points(sfm);

  lap("fundamental matrices");
  eg3d_ctx* ctx = nullptr;
  if (eg3d_create(&sc, 0, &ctx) != EG3D_OK) return fail("eg3d_create");
  lap("eg3d_create (incl. HIP runtime start-up)");
  Cloud all_stages;
  eg3d_stage_times tm;
  // ---- pipelines 1 and 2 (pipelines.cpp:219-223): the extractor over the polyline matches of each stage, in
  // match order (one call takes all matches of a stage: eg3d_match_polyline_sets emits them set by set)
  for (int stage = 0; stage < 2; stage++) {
    if (!sets_path[stage]) continue;
    MatchSets ms;
    if (!read_match_sets(sets_path[stage], V, ms)) return fail("reading a polyline match file");
    eg3d_polyline_sets ps;
    ps.n_sets = ms.n_sets;
    ps.row_off = ms.row_off.data();
    ps.pl_ids = ms.ids.data();
    eg3d_edgepoints e;
    if (eg3d_match_polyline_sets(ctx, &ps, 0, ms.n_sets, 0, &e, &tm) != EG3D_OK) return fail("eg3d_match_polyline_sets");
    std::printf("pipeline %d: %u polyline matches -> %llu edge-points (%llu observations) in %.2f ms on the GPU\n", stage + 1,
                ms.n_sets, (unsigned long long)e.n_points, (unsigned long long)e.n_obs, tm.ms_total);
    all_stages.append(e);
    eg3d_free_edgepoints(&e);
  }
  // ---- pipeline 3 on the GPU (pipelines.cpp:160-176, :227)
  eg3d_seeds seeds;
  eg3d_sfm_seeds(sfm, &seeds);
  {
    eg3d_edgepoints e;
    if (eg3d_match_refpoints(ctx, &seeds, 0, seeds.n_seeds, 0, &e, &tm) != EG3D_OK) return fail("eg3d_match_refpoints");
    std::printf("matched %u reference points -> %llu edge-points (%llu observations) in %.2f ms on the GPU\n", seeds.n_seeds,
                (unsigned long long)e.n_points, (unsigned long long)e.n_obs, tm.ms_total);
    lap("eg3d_match_refpoints (first call, with the copy)");
    all_stages.append(e);
    eg3d_free_edgepoints(&e);
    lap("append to the run's cloud");
  }
  eg3d_edgepoints pts = all_stages.view_as_edgepoints();

  // ---- filter_3d_points_close_2d_array + add_3dpoints_to_sfmd (pipelines.cpp:236-239; edge_matcher.cpp:150-158)
  std::vector<uint8_t> keep(pts.n_points ? pts.n_points : 1);
  if (eg3d_host_filter_close_2d(V, sc.width, sc.height, &pts, keep.data()) != 0) return fail("dedup");
  lap("3 px de-duplication (filter_3d_points_close_2d_array)");
  uint64_t kept = 0;
  for (uint64_t i = 0; i < pts.n_points; i++) kept += keep[i];
  if (eg3d_sfm_add_edgepoints(sfm, &pts, keep.data()) != 0) return fail("adding the edge-points");
  lap("add_3dpoints_to_sfmd");
  std::printf("kept %llu edge-points after the 3 px de-duplication; SfM data now holds %llu points\n",
              (unsigned long long)kept, (unsigned long long)eg3d_sfm_n_points(sfm));

  // ---- ./filter -e (src/utils/filter.cpp:48-115): Gauss-Newton refinement + observation-count filter
  if (do_filter) {
    eg3d_seeds all;
    eg3d_sfm_seeds(sfm, &all);
    const uint64_t n = all.n_seeds;
    std::vector<float> Xo(3 * n);
    std::vector<uint8_t> inl(n ? n : 1);
    if (eg3d_gn_filter(ctx, eg3d_sfm_points(sfm), all.trk_off, all.trk_view, all.trk_xy, n, 2.25f, 0, Xo.data(),
                       inl.data(), nullptr) != EG3D_OK)
      return fail("eg3d_gn_filter");
    eg3d_sfm_set_point_coords(sfm, Xo.data());   // inliers moved, outliers returned unchanged
    const int thr = eg3d_host_observation_filter(V, all.trk_off, n, first_edgepoint, -1, inl.data());
    uint64_t n_in = 0;
    for (uint64_t i = 0; i < n; i++) n_in += inl[i];
    eg3d_sfm_remove_outliers(sfm, inl.data());
    std::printf("filter: %llu of %llu points kept (Gauss-Newton mse < 2.25, more than %d observations)\n",
                (unsigned long long)n_in, (unsigned long long)n, thr);
    lap("filter (Gauss-Newton on the GPU + observation count)");
  }

  // ---- output_sfm_data (edge_matcher.cpp:169)
  if (eg3d_sfm_write_json(sfm, argv[1], argv[3]) != 0) return fail("writing the output");
  lap("write the OpenMVG JSON");
  std::printf("wrote %s (%llu points)\n", argv[3], (unsigned long long)eg3d_sfm_n_points(sfm));
  if (print_times)
    std::fprintf(stderr, "[time] %-44s %9.2f ms\n", "whole run", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  eg3d_destroy(ctx);
  eg3d_plg_destroy(plg);
  eg3d_sfm_destroy(sfm);
  return 0;
}
