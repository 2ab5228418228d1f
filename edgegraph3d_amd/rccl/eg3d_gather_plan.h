// eg3d_gather_plan.h — the host-side arithmetic of the cloud exchange (include/eg3d_rccl.h), shared by the
// RCCL library (device buffers) and by libeg3d_host.so (host buffers: eg3d_host_gather_plan /
// eg3d_host_gather_place, which the world_size-2 gloo test drives on CPU). Rank order = seed order
// (plg_matching_from_refpoints.cpp:83-104 has no cross-seed state), so the gathered cloud is the ranks'
// clouds back to back; a rank's observation offsets are rebased by the observations of the ranks before it.
#pragma once
#include <stdint.h>

namespace eg3d_gather_detail {

// counts3[r] = {n_points, n_obs, status}. Fills the per-rank bases and the totals; returns false when any
// rank reported a non-zero status (its local result is missing or incomplete).
inline bool plan(int n_ranks, const uint64_t* counts3, uint64_t* point_base, uint64_t* obs_base, uint64_t* total_points,
                 uint64_t* total_obs) {
  uint64_t tp = 0, to = 0, bad = 0;
  for (int r = 0; r < n_ranks; r++) {
    if (point_base) point_base[r] = tp;
    if (obs_base) obs_base[r] = to;
    tp += counts3[3 * r];
    to += counts3[3 * r + 1];
    bad |= counts3[3 * r + 2];
  }
  if (total_points) *total_points = tp;
  if (total_obs) *total_obs = to;
  return bad == 0;
}

// bytes per element of the seven fields of a cloud, and whether a field is per point (else per observation)
static const uint64_t kFieldBytes[7] = {12, 8, 16, 4, 4, 4, 8};  // X, obs_off, key | obs_view, obs_pl, obs_seg, obs_xy
inline bool field_per_point(int f) { return f < 3; }

}  // namespace eg3d_gather_detail
