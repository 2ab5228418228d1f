// Host steps after the hot path: the order-dependent 3 px occupancy-grid dedup
// and the observation-count filter. Behaviour reproduced: filter_3d_points_close_2d_array
// (reference src/edgegraph3d/filtering/filtering_close_plgps.cpp:75-124) and the tail of
// compute_inliers / compute_ray_stats (src/edgegraph3d/filtering/outliers_filtering.cpp:14-64).
// Own design: one byte-map per view in a single allocation, greedy pass over the SoA output.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/eg3d_host.h"

extern "C" int eg3d_host_filter_close_2d(int n_views, int width, int height, const eg3d_edgepoints* pts,
                                         uint8_t* keep) {
  if (!pts || !keep || n_views <= 0) return -1;
  const int CELL = 3;
  const int w = (int)std::ceil((float)width / CELL), h = (int)std::ceil((float)height / CELL);
  const size_t plane = (size_t)w * h;
  std::vector<uint8_t> occ(plane * (size_t)n_views, 0);
  // cell of an observation, or -1 when it lies outside the occupancy maps (coordinates at or past
  // the image border, negative, NaN, or a view id out of range). The reference indexes its arrays
  // unchecked there (undefined behaviour); here such an observation never makes a point "fresh"
  // and marks nothing — the CPU oracle adopts the same rule.
  auto cell_of = [&](uint64_t j) -> int64_t {
    const int32_t v = pts->obs_view[j];
    const float fx = pts->obs_xy[2 * j] / CELL, fy = pts->obs_xy[2 * j + 1] / CELL;
    if (v < 0 || v >= n_views || !(fx > -1.0f) || !(fy > -1.0f) || !(fx < (float)w) || !(fy < (float)h)) return -1;
    const int cx = (int)fx, cy = (int)fy;
    if (cx < 0 || cy < 0 || cx >= w || cy >= h) return -1;
    return (int64_t)(plane * (size_t)v + (size_t)cy * w + cx);
  };
  for (uint64_t i = 0; i < pts->n_points; i++) {
    const uint64_t a = pts->obs_off[i], b = pts->obs_off[i + 1];
    bool fresh = false;
    for (uint64_t j = a; j < b && !fresh; j++) {
      const int64_t c = cell_of(j);
      fresh = c >= 0 && occ[(size_t)c] == 0;
    }
    keep[i] = fresh;
    if (fresh)
      for (uint64_t j = a; j < b; j++) {
        const int64_t c = cell_of(j);
        if (c >= 0) occ[(size_t)c] = 1;
      }
  }
  return 0;
}

extern "C" int eg3d_host_observation_filter(int n_cameras, const uint32_t* obs_off, uint64_t n_points,
                                            uint64_t first_edgepoint, int forced_min_filter, uint8_t* inlier) {
  std::vector<uint64_t> hist((size_t)n_cameras + 1, 0);
  uint64_t count = 0;
  for (uint64_t i = 0; i < n_points; i++)
    if (inlier[i]) {
      count++;
      uint32_t k = obs_off[i + 1] - obs_off[i];
      if (k >= 1 && k <= (uint32_t)n_cameras) hist[k - 1]++;
    }
  uint64_t acc = 0;
  int median = 0;
  for (median = 0; median < n_cameras; median++) {
    acc += hist[median];
    if (acc >= count / 2) break;
  }
  int threshold = median / 2 - 1;
  if (threshold < 3) threshold = 3;  // FILTER_3VIEWS_AMOUNT
  if (forced_min_filter > -1) threshold = forced_min_filter;
  for (uint64_t i = first_edgepoint; i < n_points; i++)
    if (inlier[i] && !((int)(obs_off[i + 1] - obs_off[i]) > threshold)) inlier[i] = 0;
  return threshold;
}
