// gather_host.cpp — the placement arithmetic of the cloud exchange on host arrays (include/eg3d_host.h:
// eg3d_host_gather_plan / eg3d_host_gather_place). The RCCL library (edgegraph3d_amd/rccl/eg3d_rccl.hip) uses the
// same plan on device buffers; this is what a CPU transport calls around its own all-gather of the raw arrays
// (tests/test_multirank_gloo.py, world_size 2).
#include <stdint.h>
#include <string.h>

#include "../../include/eg3d_host.h"
#include "../rccl/eg3d_gather_plan.h"

extern "C" int eg3d_host_gather_plan(int n_ranks, const uint64_t* counts3, uint64_t* point_base, uint64_t* obs_base,
                                     uint64_t* total_points, uint64_t* total_obs) {
  if (n_ranks < 1 || !counts3) return -1;
  return eg3d_gather_detail::plan(n_ranks, counts3, point_base, obs_base, total_points, total_obs) ? 0 : -4;
}

extern "C" int eg3d_host_gather_place(const eg3d_edgepoints* part, uint64_t point_base, uint64_t obs_base,
                                      eg3d_edgepoints* whole) {
  if (!part || !whole) return -1;
  const uint64_t np = part->n_points, no = part->n_obs;
  if (point_base + np > whole->n_points || obs_base + no > whole->n_obs) return -1;
  if (np && (!part->X || !part->obs_off || !part->key || !whole->X || !whole->obs_off || !whole->key)) return -1;
  if (no && (!part->obs_view || !part->obs_pl || !part->obs_seg || !part->obs_xy || !whole->obs_view || !whole->obs_pl ||
             !whole->obs_seg || !whole->obs_xy))
    return -1;
  if (np) {
    memcpy(whole->X + 3 * point_base, part->X, sizeof(float) * 3 * np);
    memcpy(whole->key + 4 * point_base, part->key, sizeof(uint32_t) * 4 * np);
    for (uint64_t i = 0; i < np; i++) whole->obs_off[point_base + i] = part->obs_off[i] + obs_base;
  }
  if (no) {
    memcpy(whole->obs_view + obs_base, part->obs_view, sizeof(int32_t) * no);
    memcpy(whole->obs_pl + obs_base, part->obs_pl, sizeof(uint32_t) * no);
    memcpy(whole->obs_seg + obs_base, part->obs_seg, sizeof(uint32_t) * no);
    memcpy(whole->obs_xy + 2 * obs_base, part->obs_xy, sizeof(float) * 2 * no);
  }
  return 0;
}
