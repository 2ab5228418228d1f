// eg3d_kernels.hip — gfx950 kernels of the refpoint -> epipolar match -> triangulate path.
//
// Phase pipeline (DESIGN.md "Kernels"):
//   k_seed_prep       1 lane / seed       view map of the track, (seed,entry) of every track entry
//   k1_count_raw      1 lane / (seed,entry)  upper bound of candidate polylines (sizes the slots)
//   k1_seed_candidates 1 WAVE / (seed,entry) k-way merge of the 30 px grid cells into batches of 64 candidate ids +
//                                           closest-point scan of a batch's segments as one flat sequence
//   k_task_fill       1 lane / (seed,entry)  enumerates (seed, start view, start hit) tasks; sums K1's byte counts
//   k2_epipolar_hits  1 WAVE / task         epiline x the segments of all candidate polylines of a list (flat),
//                                           ballot/popcount ordered compaction (count pass + fill pass)
//   k_task_setup      1 lane / task         3-view selection, hypothesis count
//   k3a_orient, k3a_follow_spec  1 lane / hypothesis, 1 lane / list — the wave SERVES its lanes' triangulation requests
//                                           densely from 64 slots in LDS (eg3d_k3a_engine.h)
//   k3s_select        1 lane / task         uniqueness rule -> chain seeds
//   k3b_expand        1 WAVE / chain        expand-all-views (wave-cooperative Gauss-Newton)
//   k4_emit           1 WAVE / chain        ordered SoA output (wave prefix sum of obs counts, flat coalesced copy)
//   k5_gn_filter      1 lane / point        config 5, FP32 Gauss-Newton outlier filter
// Pipelines 1-2 extractor, stage A' feeding the same task_setup..k4 stages:
//   k_n1_samples      1 lane / polyline     a sample every 20 px (count pass + fill pass), 1 task each
//   k_n1_hits         1 lane / task        epiline x the set's polylines of every view (wave-uniform scan), all hits
// All arithmetic follows the contract in DESIGN.md (no FMA contraction: -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eg3d_dev_pipeline.h"
#include "eg3d_dev_coopgn.h"
#include "eg3d_kernels.h"

namespace eg3d {

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t t = (uint32_t)__shfl_xor((int)v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}
__device__ __forceinline__ float wave_min_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float t = __shfl_xor(v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

// Wave minimum through DPP (row_shr 1/2/4/8, row_bcast 15/31: lane 63 ends up with the minimum of all lanes) and one
// v_readlane — seven short instructions instead of six dependent trips through the LDS crossbar. All lanes active.
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
  auto mn = [](uint32_t x, int y) { return (uint32_t)y < x ? (uint32_t)y : x; };
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
  v = mn(v, __builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
  return (uint32_t)lane_bcast((int)v, 63);
}

// ------------------------------------------------------------------ prep -------
__global__ void k_seed_prep(SeedsDev sd, uint32_t seed_begin, uint32_t n_seeds, uint32_t sv_base, uint32_t* sv_seed,
                            int32_t* map_view, uint32_t* map_entry, uint32_t* map_n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_seeds) return;
  uint32_t seed = seed_begin + i;
  uint32_t t0 = sd.trk_off[seed], t1 = sd.trk_off[seed + 1];
  for (uint32_t e = t0; e < t1; e++) sv_seed[e - sv_base] = seed;
  map_n[i] = build_seed_view_map(sd.trk_view + t0, t1 - t0, map_view + (t0 - sv_base), map_entry + (t0 - sv_base));
}

// Observation of `seed` in view `view`: the LAST track entry with that view id (Q2).
__device__ __forceinline__ void seed_obs_in_view(const SeedsDev& sd, uint32_t t0, uint32_t k, int32_t view, float& x,
                                                 float& y) {
  x = 0.f;
  y = 0.f;
  for (uint32_t i = 0; i < k; i++)
    if (sd.trk_view[t0 + i] == view) {
      x = sd.trk_xy[2 * (t0 + i)];
      y = sd.trk_xy[2 * (t0 + i) + 1];
    }
}

__global__ void k1_count_raw(DevScene s, SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
                             uint32_t* raw_cnt) {
  uint32_t sv = blockIdx.x * blockDim.x + threadIdx.x;
  if (sv >= n_sv) return;
  uint32_t seed = sv_seed[sv];
  uint32_t t0 = sd.trk_off[seed], k = sd.trk_off[seed + 1] - t0;
  int32_t view = sd.trk_view[sv_base + sv];
  float px, py;
  seed_obs_in_view(sd, t0, k, view, px, py);
  CellWindow w = cell_window(30.0f, s.width, s.height, s.g30_w, s.g30_h, px, py);
  uint32_t n = 0;
  if (w.c1 >= w.c0) {
    const size_t base = (size_t)view * (size_t)(s.g30_w * s.g30_h);
    for (int r = w.r0; r <= w.r1; r++)
      n += s.g30_off[base + (size_t)r * s.g30_w + w.c1 + 1] - s.g30_off[base + (size_t)r * s.g30_w + w.c0];
  }
  raw_cnt[sv] = n;
}

// ------------------------------------------------------------------ K0 ---------
// Uniform-grid construction on the device (round 6). Behaviour reproduced: PolyLine2DMap ctor +
// polyline::get_intersectedcells_2dmap_set (matching/plg_matching/polyLine_2d_map.cpp:40-58, plgs/polyline_graph_2d.cpp:
// 555-577,819-835): every polyline is sampled from its start every cell / (1.414 + 0.1) px (Euclidean stepping: the walk of
// eg3d_dev_geom.h, the one the kernels use everywhere), samples on a cell boundary are dropped, and each remaining sample's
// cell lists the polyline once. Own design: one lane walks one polyline and emits 64-bit keys (view, cell, polyline) — first
// counted, then written behind an exclusive scan —, a radix sort + unique of the keys IS the per-cell ascending id list,
// and one pass over the unique keys writes the CSR offsets. (host/grid_build.cpp is the same statement for one view on the
// host: the tests compare the two, and both with the oracle.)
#define EG3D_K0_PL_BITS 19 /* a view holds <= 524 288 polylines (eg3d_create refuses more) */
static_assert(EG3D_K0_PL_BITS == EG3D_K0_PL_BITS_HOST, "key layout of the grid builder");
template <bool FILL>
__global__ void k0_grid_pairs(DevScene s, uint32_t n_pl, float cell_dim, int map_w, int map_h, uint32_t* cnt, const uint32_t* off,
                              unsigned long long* keys, uint32_t* dropped) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_pl) return;
  const uint32_t a = s.pl_vtx_off[g], b = s.pl_vtx_off[g + 1];
  uint32_t count = 0;
  if (b - a >= 2) {  // (an invalidated polyline has no vertices on the device)
    uint32_t lo = 0, hi = (uint32_t)s.n_views;  // view of polyline g: last v with view_pl_off[v] <= g
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (s.view_pl_off[mid] <= g) lo = mid; else hi = mid;
    }
    const unsigned long long cell_base = (unsigned long long)lo * (unsigned long long)((uint32_t)map_w * (uint32_t)map_h);
    const unsigned long long pl_local = g - s.view_pl_off[lo];
    PlRef pl;
    pl.v = s.vtx + a;
    pl.n = b - a;
    pl.start = s.pl_start[g];
    pl.end = s.pl_end[g];
    const float step = (float)(cell_dim / (1.414 + 0.1));
    const uint32_t direction = pl.end;  // get_other_end(start): end, which equals start for loops (the walk stops at once: Q8)
    const uint32_t base = FILL ? off[g] : 0u;
    bool have_prev = false, pushed = false;
    int32_t prev_c = 0, prev_r = 0;
    uint32_t n_dropped = 0;
    auto visit = [&](const PlPt& p) {
      const CellCoord cc = cell_of(cell_dim, p.x, p.y);
      if (cc.bx || cc.by) return;
      if (have_prev && pushed && cc.col == prev_c && cc.row == prev_r) return;
      if (cc.col < 0 || cc.col >= map_w || cc.row < 0 || cc.row >= map_h) {
        n_dropped++;  // the reference indexes out of bounds here; inputs must keep vertices inside the image
      } else {
        if (FILL) keys[base + count] = ((cell_base + (unsigned long long)(cc.row * map_w + cc.col)) << EG3D_K0_PL_BITS) | pl_local;
        count++;
        pushed = true;
      }
      prev_c = cc.col;
      prev_r = cc.row;
      have_prev = true;
    };
    PlPt cur;
    cur.seg = 0;
    cur.x = pl.v[0].x;
    cur.y = pl.v[0].y;
    visit(cur);
    for (;;) {
      PlPt nx;
      const uint32_t w = walk_by_distance(pl, cur, direction, step, nx);
      visit(nx);
      cur = nx;
      if (w & WALK_EXTREME) break;
    }
    if (!FILL && n_dropped) atomicAdd(dropped, n_dropped);
  }
  if (!FILL) cnt[g] = count;
}
// unique sorted keys -> CSR over (view, cell): off[c] = first key whose cell is >= c, ids = the polyline ids
__global__ void k0_grid_csr(const unsigned long long* keys, uint32_t n, uint32_t total_cells, uint32_t* off, uint32_t* ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const uint32_t cell = (uint32_t)(k >> EG3D_K0_PL_BITS);
  ids[i] = (uint32_t)(k & ((1ull << EG3D_K0_PL_BITS) - 1ull));
  const uint32_t first = i ? (uint32_t)(keys[i - 1] >> EG3D_K0_PL_BITS) + 1u : 0u;
  for (uint32_t c = first; c <= cell; c++) off[c] = i;
  if (i == n - 1)
    for (uint32_t c = cell + 1; c <= total_cells; c++) off[c] = n;
}

// ------------------------------------------------------------------ K1 ---------
// One wavefront per (seed, track entry). Lanes 0..8 each own one grid cell of the (shrunk)
// 3x3 window and k-way-merge the ascending id lists (wave-min of the heads) so candidates
// come out ascending and unique, 64 at a time; the segments of such a batch are scanned as one
// flat sequence over the 64 lanes and the first closest segment of every candidate falls out of
// an LDS minimum (below). Outputs go to the slot [raw_off[sv], raw_off[sv+1]) sized by
// k1_count_raw. (Reference: PLGEdgeManager::detect_nearby_intersections_and_correspondences_plgp,
// plg_edge_manager.cpp:261-300, its per-view polyline search polyLine_2d_map_search.cpp:46-77.)
__global__ void __launch_bounds__(256) k1_seed_candidates(DevScene s, SeedsDev sd, uint32_t sv_base, uint32_t n_sv,
                                                         const uint32_t* sv_seed, const uint32_t* raw_off,
                                                         uint32_t* cand_pl, Obs* start_hits, uint32_t* cand_cnt,
                                                         uint32_t* start_cnt, uint32_t* sv_vtx) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (wave >= n_sv) return;
  const uint32_t sv = wave;
  const uint32_t seed = sv_seed[sv];
  const uint32_t t0 = sd.trk_off[seed], k = sd.trk_off[seed + 1] - t0;
  const int32_t view = sd.trk_view[sv_base + sv];
  float px, py;
  seed_obs_in_view(sd, t0, k, view, px, py);
  CellWindow w = cell_window(30.0f, s.width, s.height, s.g30_w, s.g30_h, px, py);
  uint32_t a = 0, b = 0;
  if (w.c1 >= w.c0) {
    const int ncols = w.c1 - w.c0 + 1, nrows = w.r1 - w.r0 + 1;
    if ((int)lane < ncols * nrows) {
      const int r = w.r0 + (int)lane / ncols, c = w.c0 + (int)lane % ncols;
      const size_t cell = (size_t)view * (size_t)(s.g30_w * s.g30_h) + (size_t)r * s.g30_w + c;
      a = s.g30_off[cell];
      b = s.g30_off[cell + 1];
    }
  }
  const uint32_t out_base = raw_off[sv];
  uint32_t nc = 0, ns = 0;
  uint32_t nvtx = 0;
  // Candidates are taken in batches of up to 64 ids (phase A: the k-way merge, one wave-minimum per id, the lanes' list
  // heads loaded eight at a time); the segments of a whole batch are then scanned as ONE flat sequence (phase B) — a
  // pass per candidate paid the dependent look-ups id -> vertex range -> vertices and three wave reductions per
  // candidate with ~25 of 64 lanes busy. The first closest segment of a candidate (smallest distance, then smallest
  // index) is the minimum of the 64-bit keys (distance bits : segment) in the candidate's LDS slot: squared distances
  // are >= +0, so their bit patterns order like the values; NaN / infinite distances are never submitted (the plain
  // scan's `d < best` with best = +inf).
  __shared__ unsigned long long k1_best[4][64];
  unsigned long long* const slot = k1_best[threadIdx.x >> 6];
  const uint32_t gview = s.view_pl_off[view];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint32_t h0 = 0xffffffffu, h1 = 0xffffffffu, h2 = 0xffffffffu, h3 = 0xffffffffu, h4 = 0xffffffffu, h5 = 0xffffffffu,
           h6 = 0xffffffffu, h7 = 0xffffffffu;
  uint32_t hn = 0;  // ids of this lane's list held in h0..h7
  for (;;) {
    uint32_t my_id = 0xffffffffu, nb = 0;
    while (nb < 64) {
      if (hn == 0 && a < b) {  // refill: up to eight ids of the lane's cell list, their loads in flight together
        const uint32_t r = b - a;
        h0 = s.g30_ids[a];
        h1 = r > 1 ? s.g30_ids[a + 1] : 0xffffffffu;
        h2 = r > 2 ? s.g30_ids[a + 2] : 0xffffffffu;
        h3 = r > 3 ? s.g30_ids[a + 3] : 0xffffffffu;
        h4 = r > 4 ? s.g30_ids[a + 4] : 0xffffffffu;
        h5 = r > 5 ? s.g30_ids[a + 5] : 0xffffffffu;
        h6 = r > 6 ? s.g30_ids[a + 6] : 0xffffffffu;
        h7 = r > 7 ? s.g30_ids[a + 7] : 0xffffffffu;
        hn = r > 8 ? 8 : r;
        a += hn;
      }
      const uint32_t head = hn ? h0 : 0xffffffffu;
      const uint32_t m = wave_min_u32_dpp(head);
      if (m == 0xffffffffu) break;
      if (head == m) {
        h0 = h1, h1 = h2, h2 = h3, h3 = h4, h4 = h5, h5 = h6, h6 = h7, h7 = 0xffffffffu;
        hn--;
      }
      if (lane == nb) my_id = m;
      nb++;
    }
    if (nb == 0) break;
    uint32_t my_a = 0, my_n = 0;
    if (lane < nb) {
      const uint32_t v0 = s.pl_vtx_off[gview + my_id], v1 = s.pl_vtx_off[gview + my_id + 1];
      my_a = v0;
      my_n = v1 - v0;
    }
    const uint32_t my_ns = my_n >= 2u ? my_n - 1u : 0u;
    nvtx += (uint32_t)lane_bcast(wave_incl_scan((int)my_n), 63);
    const uint32_t incl = (uint32_t)wave_incl_scan((int)my_ns);
    const uint32_t total = (uint32_t)lane_bcast((int)incl, 63);
    slot[lane] = ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base = 0; base < total; base += 64) {
      const uint32_t f = base + lane;
      uint32_t pos = 0;  // owner of flat segment f: the first lane whose inclusive count exceeds f
#pragma unroll
      for (uint32_t step = 32; step; step >>= 1) {
        const uint32_t v = (uint32_t)__shfl((int)incl, (int)(pos + step - 1), 64);
        if (v <= f) pos += step;
      }
      const uint32_t o_incl = (uint32_t)__shfl((int)incl, (int)pos, 64);
      const uint32_t o_ns = (uint32_t)__shfl((int)my_ns, (int)pos, 64);
      const uint32_t o_a = (uint32_t)__shfl((int)my_a, (int)pos, 64);
      if (f < total) {
        const uint32_t j = f - (o_incl - o_ns);
        const f2 v0 = s.vtx[o_a + j], v1 = s.vtx[o_a + j + 1];
        float qx, qy;
        const float d = seg_closest(px, py, v0.x, v0.y, v1.x, v1.y, qx, qy);
        if (d < __builtin_huge_valf())
          atomicMin(&slot[pos], ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)j);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long key = slot[lane];
    const bool valid = lane < nb && key != ~0ull;  // no finite distance (degenerate polyline): never a candidate
    const float dmin = __uint_as_float((uint32_t)(key >> 32));
    const uint32_t jmin = (uint32_t)key;
    const bool is_start = valid && dmin <= 100.0f;
    const bool is_cand = valid && dmin <= 900.0f;
    const unsigned long long mc = __ballot(is_cand), ms = __ballot(is_start);
    if (is_cand) cand_pl[out_base + nc + __popcll(mc & lt_mask)] = my_id;
    if (is_start) {
      const f2 v0 = s.vtx[my_a + jmin], v1 = s.vtx[my_a + jmin + 1];
      Obs o;
      o.view = view;
      o.pl = my_id;
      o.seg = jmin;
      (void)seg_closest(px, py, v0.x, v0.y, v1.x, v1.y, o.x, o.y);  // the winning segment's closest point again: same inputs, same bits
      start_hits[out_base + ns + __popcll(ms & lt_mask)] = o;
    }
    nc += __popcll(mc);
    ns += __popcll(ms);
    __builtin_amdgcn_wave_barrier();
    if (nb < 64) break;
  }
  if (lane == 0) {
    cand_cnt[sv] = nc;
    start_cnt[sv] = ns;
    // the vertices this entry's scans touched (algorithmic bytes): summed by k_task_fill, one atomic per
    // wave of entries — one atomic per entry on the one counter serialised at the memory side and was 3/4 of this
    // kernel's time (0.51 -> 0.13 ms on C3')
    sv_vtx[sv] = nvtx;
  }
}

__global__ void k_task_fill(SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
                            const uint32_t* start_cnt, const uint32_t* task_off, uint32_t* task_seed,
                            uint32_t* task_entry, uint32_t* task_hit, uint32_t* task_k, const uint32_t* sv_vtx,
                            Counters* ctr) {
  uint32_t sv = blockIdx.x * blockDim.x + threadIdx.x;
  {  // K1's vertex counts -> the byte counter, one atomic per wave of entries
    unsigned long long v = sv < n_sv ? (unsigned long long)sv_vtx[sv] : 0ull;
    for (int d = 32; d; d >>= 1) v += (unsigned long long)__shfl_xor((long long)v, d, 64);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&ctr->bytes, 8ull * v);
  }
  if (sv >= n_sv) return;
  const uint32_t seed = sv_seed[sv];
  const uint32_t t0 = sd.trk_off[seed], k = sd.trk_off[seed + 1] - t0;
  const uint32_t entry = sv_base + sv - t0;
  const uint32_t n = start_cnt[sv], base = task_off[sv];
  for (uint32_t h = 0; h < n; h++) {
    task_seed[base + h] = seed;
    task_entry[base + h] = entry;
    task_hit[base + h] = h;
    task_k[base + h] = k;
  }
}

// ------------------------------------------------------------------ K2 ---------
// One wavefront per task. For every other track entry: epipolar line of the start hit, then
// all 64 lanes test consecutive segments of each candidate polyline; hits inside the
// detection radius are compacted in segment order with __ballot + popcount. FILL=false
// counts, FILL=true writes to the offsets produced by the scan of the counts.
template <bool FILL>
__global__ void __launch_bounds__(256) k2_epipolar_hits(DevScene s, SeedsDev sd, uint32_t sv_base, uint32_t n_tasks,
                                                       const uint32_t* task_seed, const uint32_t* task_entry,
                                                       const uint32_t* task_hit, const uint32_t* task_list_off,
                                                       const uint32_t* raw_off, const uint32_t* cand_pl,
                                                       const uint32_t* cand_cnt, const Obs* start_hits,
                                                       uint32_t* list_cnt, const uint32_t* list_ptr, Obs* hits) {
  const uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (t >= n_tasks) return;
  const uint32_t seed = task_seed[t], ea = task_entry[t], h = task_hit[t];
  const uint32_t t0 = sd.trk_off[seed], k = sd.trk_off[seed + 1] - t0;
  const uint32_t sv0 = t0 - sv_base;
  const Obs hit = start_hits[raw_off[sv0 + ea] + h];
  const int32_t start_view = sd.trk_view[t0 + ea];
  float ix, iy;
  seed_obs_in_view(sd, t0, k, start_view, ix, iy);
  const float radius = dist(ix, iy, hit.x, hit.y) * 3.0f;
  const float detsq = radius * radius;
  const uint32_t lo = task_list_off[t];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (uint32_t i = 0; i < k; i++) {
    const int32_t cur_view = sd.trk_view[t0 + i];
    uint32_t cnt = 0;
    if (cur_view == start_view) {
      cnt = 1;
      if (FILL && lane == 0) {
        Obs o = hit;
        o.view = cur_view;
        hits[list_ptr[lo + i]] = o;
      }
    } else {
      float la, lb, lc;
      if (epiline(s.F, s.F_valid, s.n_views, start_view, cur_view, hit.x, hit.y, la, lb, lc)) {
        const float sx = sd.trk_xy[2 * (t0 + i)], sy = sd.trk_xy[2 * (t0 + i) + 1];
        const uint32_t cbase = raw_off[sv0 + i], ncand = cand_cnt[sv0 + i];
        const uint32_t wbase = FILL ? list_ptr[lo + i] : 0;
        if (FILL && list_ptr[lo + i + 1] == wbase) continue;  // the count pass found nothing for this list
        // The segments of up to 64 candidate polylines are dealt to the lanes as ONE flat sequence (candidate-major,
        // segment-minor = the order of the per-candidate loops): polylines average ~25 vertices, so a pass per
        // candidate left 60 % of the lanes idle and paid its dependent look-ups (candidate id -> vertex range ->
        // vertices) once per candidate; here the look-ups of all candidates are in flight together.
        const uint32_t gview = s.view_pl_off[cur_view];
        for (uint32_t c0 = 0; c0 < ncand; c0 += 64) {
          const uint32_t c = c0 + lane;
          uint32_t my_id = 0, my_a = 0, my_ns = 0;
          if (c < ncand) {
            my_id = cand_pl[cbase + c];
            const uint32_t a = s.pl_vtx_off[gview + my_id], b = s.pl_vtx_off[gview + my_id + 1];
            my_a = a;
            my_ns = (b - a) >= 2u ? (b - a) - 1u : 0u;
          }
          const uint32_t incl = (uint32_t)wave_incl_scan((int)my_ns);
          const uint32_t total = (uint32_t)lane_bcast((int)incl, 63);
          const uint32_t excl = incl - my_ns;
          // four chunks of 64 flat segments per trip: their owner searches and vertex loads are independent and in
          // flight together (one chunk at a time, a wave waited out one memory latency per chunk)
          constexpr int U = 4;
          for (uint32_t base = 0; base < total; base += 64 * U) {
            f2 v0[U], v1[U];
            uint32_t oid[U], seg[U];
            bool in[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
              const uint32_t f = base + 64u * u + lane;
              in[u] = f < total;
              oid[u] = seg[u] = 0;
              v0[u].x = v0[u].y = v1[u].x = v1[u].y = 0.f;
              if (base + 64u * u < total) {  // wave-uniform
                uint32_t pos = 0;  // owner of flat segment f: the first lane whose inclusive count exceeds f
#pragma unroll
                for (uint32_t step = 32; step; step >>= 1) {
                  const uint32_t v = (uint32_t)__shfl((int)incl, (int)(pos + step - 1), 64);
                  if (v <= f) pos += step;
                }
                const uint32_t o_excl = (uint32_t)__shfl((int)excl, (int)pos, 64);
                const uint32_t o_a = (uint32_t)__shfl((int)my_a, (int)pos, 64);
                oid[u] = (uint32_t)__shfl((int)my_id, (int)pos, 64);
                if (in[u]) {
                  seg[u] = f - o_excl;
                  v0[u] = s.vtx[o_a + seg[u]];
                  v1[u] = s.vtx[o_a + seg[u] + 1];
                }
              }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
              if (base + 64u * u >= total) break;  // wave-uniform
              bool ok = false;
              float hx = 0.f, hy = 0.f;
              if (in[u] && seg_line_hit(v1[u].x, v1[u].y, v0[u].x, v0[u].y, la, lb, lc, hx, hy))
                ok = dist2(sx, sy, hx, hy) <= detsq;
              const unsigned long long mask = __ballot(ok);
              if (FILL && ok) {
                Obs o;
                o.view = cur_view;
                o.pl = oid[u];
                o.seg = seg[u];
                o.x = hx;
                o.y = hy;
                hits[wbase + cnt + __popcll(mask & lt_mask)] = o;
              }
              cnt += __popcll(mask);
            }
          }
        }
      }
    }
    if (!FILL && lane == 0) list_cnt[lo + i] = cnt;
  }
}


// ------------------------------------------------------------------ N1 ---------
// Pipelines 1-2 extractor, stage A (polyline_matching.cpp:153-208 with :45-73): every polyline of a
// set is sampled every 20 px from its start towards its end; each sample is one task whose V lists
// are the hits of its epipolar line on the set's polylines of the other views (all hits, no
// radius; the list of its own view is the sample itself).
__device__ __forceinline__ uint32_t n1_row_of_item(const uint32_t* row_off, uint32_t n_rows, uint32_t item) {
  uint32_t lo = 0, hi = n_rows;  // largest row with row_off[row] <= item (empty rows skipped by <=)
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (row_off[mid] <= item)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}
template <bool FILL>
__global__ void k_n1_samples(DevScene s, SetsDev sets, uint32_t n_rows, uint32_t item_begin, uint32_t n_items,
                             uint32_t* sample_cnt, const uint32_t* sample_off, Obs* samples, uint32_t* task_seed,
                             uint32_t* task_entry, uint32_t* task_hit, uint32_t* task_list_off, uint32_t* task_row0,
                             Counters* ctr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  const uint32_t item = item_begin + i;
  const uint32_t row = n1_row_of_item(sets.row_off, n_rows, item);
  const int view = (int)(row % sets.n_views);
  const uint32_t pl_id = sets.pl_ids[item];
  const PlRef pl = polyline_of(s, view, pl_id);
  uint32_t n = 0;
  if (pl.n >= 2) {
    PlPt p, q;
    p.seg = 0;
    p.x = pl.v[0].x;
    p.y = pl.v[0].y;
    const uint32_t base = FILL ? sample_off[i] : 0u;
    for (;;) {
      const uint32_t w = walk_by_distance(pl, p, pl.end, 20.0f, q);
      if (w & WALK_BAD_DIR) atomicOr(&ctr->flags, 8u);
      if (w & WALK_EXTREME) break;
      if (FILL) {
        const uint32_t t = base + n;
        Obs o;
        o.view = view;
        o.pl = pl_id;
        o.seg = q.seg;
        o.x = q.x;
        o.y = q.y;
        samples[t] = o;
        task_seed[t] = t;
        task_entry[t] = (uint32_t)view;
        task_hit[t] = 0;
        task_list_off[t] = t * sets.n_views;
        task_row0[t] = (row / sets.n_views) * sets.n_views;  // first row of the task's set
      }
      n++;
      p = q;
    }
  }
  if (!FILL) sample_cnt[i] = n;
}

// One LANE per task, 64 consecutive tasks per wavefront. Consecutive tasks are consecutive samples of the same
// polylines, so almost always the whole wave works on ONE set: its lanes are grouped by set, and for every view the
// group scans the set's polylines of that view together — polyline ids, descriptors and vertices are wave-uniform
// (scalar loads, one fetch for 64 tasks), each lane tests the segment against ITS OWN epipolar line and appends its
// hits to ITS OWN list, in polyline and segment order by construction. (Round 1-3 launched one wavefront per
// (task, view) — 64 lanes across the segments, ballot compaction — i.e. 25 waves per task that each fetch the same
// few polylines again: 55.7 ms of the 370 ms C3' sets step; this form: 3.2 ms.)
template <bool FILL>
__global__ void __launch_bounds__(256) k_n1_hits(DevScene s, SetsDev sets, uint32_t n_tasks, const Obs* samples,
                                                const uint32_t* task_row0, uint32_t* list_cnt,
                                                const uint32_t* list_ptr, Obs* hits, Counters* ctr) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool have = t < n_tasks;
  const uint32_t V = sets.n_views;
  Obs smp;
  smp.view = 0;
  smp.pl = smp.seg = 0;
  smp.x = smp.y = 0.0f;
  uint32_t row0 = 0xffffffffu;
  if (have) {
    smp = samples[t];
    row0 = task_row0[t];
  }
  unsigned long long bytes = 0;
  unsigned long long todo = __ballot(have);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t set_row0 = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)row0, leader, 64));
    const bool mine = have && row0 == set_row0;
    todo &= ~__ballot(mine);
    for (uint32_t cur_view = 0; cur_view < V; cur_view++) {
      const size_t w = (size_t)t * V + cur_view;  // list index = task * V + view
      uint32_t cnt = 0;
      bool scan = false;
      float la = 0.0f, lb = 0.0f, lc = 0.0f;
      uint32_t wbase = 0;
      if (mine) {
        if ((uint32_t)smp.view == cur_view) {
          cnt = 1;
          if (FILL) hits[list_ptr[w]] = smp;
        } else {
          scan = epiline(s.F, s.F_valid, s.n_views, (int)smp.view, (int)cur_view, smp.x, smp.y, la, lb, lc);
          if (FILL && scan) wbase = list_ptr[w];
        }
      }
      if (__any(scan)) {
        const uint32_t row = set_row0 + cur_view;
        const uint32_t c0 = sets.row_off[row], c1 = sets.row_off[row + 1];
        for (uint32_t c = c0; c < c1; c++) {
          const uint32_t pl_id = sets.pl_ids[c];
          const PlRef pl = polyline_of(s, (int)cur_view, pl_id);
          if (scan) bytes += 8ull * pl.n;
          for (uint32_t ii = 1; ii < pl.n; ii++) {
            const f2 v1 = pl.v[ii], v0 = pl.v[ii - 1];
            float hx = 0.f, hy = 0.f;
            if (scan && seg_line_hit(v1.x, v1.y, v0.x, v0.y, la, lb, lc, hx, hy)) {  // (v[i], v[i-1]), tagged i-1: Q10
              if (FILL) {
                Obs o;
                o.view = cur_view;
                o.pl = pl_id;
                o.seg = ii - 1;
                o.x = hx;
                o.y = hy;
                hits[wbase + cnt] = o;
              }
              cnt++;
            }
          }
        }
      }
      if (!FILL && mine) list_cnt[w] = cnt;
    }
  }
  if (!FILL) {  // algorithmic bytes: the vertices every (task, view) list scanned
    for (int d = 32; d; d >>= 1) bytes += (unsigned long long)__shfl_xor((long long)bytes, d, 64);
    if ((threadIdx.x & 63u) == 0 && bytes) atomicAdd(&ctr->bytes, bytes);
  }
}

// ------------------------------------------------------------------ tasks ------
__global__ void k_task_setup(StageAView a, const int32_t* map_view, const uint32_t* map_entry, const uint32_t* map_n,
                             TaskDesc* tasks, uint32_t* n_hyp) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.n_tasks) return;
  TaskDesc d;
  task_setup(a, t, map_view, map_entry, map_n, d);
  tasks[t] = d;
  n_hyp[t] = d.n_hyp;
}

__device__ __forceinline__ uint32_t find_owner(const uint32_t* off, uint32_t n, uint32_t x) {
  // largest t in [0,n) with off[t] <= x (off ascending, off[n] > x)
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= x)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------ K3a --------
// Hypothesis evaluation: eg3d_k3a_engine.h (k3a_orient, k3a_follow_spec), included below; what it computes is stated
// sequentially by evaluate_hypothesis in eg3d_dev_follow.h (the host simulation of the tests runs that).
#ifndef EG3D_K3A_WAVES
#define EG3D_K3A_WAVES 2 /* waves/SIMD the register allocation of K3a aims at: 256 VGPRs, nothing spills (at 3: 168 VGPRs,
                            86-102 spilled; same speed on C3', K3a 1.50 -> 1.30 ms on C2, half the L2<->fabric traffic) */
#endif
// compatible <=> direction 1 gave >= 2 points, or direction 2 is valid and gave >= 2
// (compatible_new_plg_point, plg_matching.cpp:1276-1287)
__global__ void k3a_finalize(uint32_t n_hyp, HypResult* res) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= n_hyp) return;
  const uint32_t st = res[h].status;
  if ((st & HYP_D1) && (res[h].n1 >= 2 || ((st & HYP_D2) && res[h].n2 >= 2))) res[h].status = st | HYP_COMPAT;
}

}  // namespace eg3d
#include "eg3d_k3a_engine.h"
namespace eg3d {

// ------------------------------------------------------------------ K3s --------
__global__ void k3s_select(uint32_t n_tasks, const uint32_t* hyp_off, const HypResult* res, ChainSeed* seeds_out,
                           uint32_t* valid) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tasks) return;
  ChainSeed cs;
  cs.task = t;
  cs.winner = 0;
  cs.pts2_src = 0xffffffffu;
  cs.n1 = 0;
  cs.n2 = 0;
  bool ok = false;
  if (hyp_off[t + 1] > hyp_off[t]) ok = select_task(res, hyp_off[t], hyp_off[t + 1], cs);
  seeds_out[t] = cs;
  valid[t] = ok ? 1u : 0u;
}
__global__ void k_compact_chains(uint32_t n_tasks, const ChainSeed* per_task, const uint32_t* valid,
                                 const uint32_t* chain_off, ChainSeed* chains) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tasks) return;
  if (valid[t]) chains[chain_off[t]] = per_task[t];
}

// ------------------------------------------------------------------ K3b --------
// One WAVEFRONT per chain (block = 64 lanes). Control flow is wave-uniform; the lane-parallel
// sections are (a) the per-view projection + 4 px grid lookup + closest point of every chain
// point and (b) the Gauss-Newton ADD solves of a side walk's candidates (see eg3d_dev_expand.h).
#ifndef EG3D_WAVE_SLOT_STEP
#define EG3D_WAVE_SLOT_STEP 0 /* measured slower: failed speculative candidates run all 30 GN iterations */
#endif
#ifndef EG3D_DLT_HOT_IN_LDS
#define EG3D_DLT_HOT_IN_LDS 1 /* the DLTs of chain following keep their matrices in LDS (1) or in registers (0); the rare 3-subset fallback's always in LDS */
#endif
#ifndef EG3D_DLT_GRP
#define EG3D_DLT_GRP 1 /* the 2-view DLTs of the expand stage on groups of 8 lanes (dlt2_grp8, eg3d_dev_coopgn.h: rows of A and V in registers, only the ordered sums through LDS) instead of one lane per DLT with its matrices in LDS (0: rounds 3-5) */
#endif
#ifndef EG3D_REDO_SKIP
#define EG3D_REDO_SKIP 1 /* the redo of a look-ahead step whose triangulation failed does not walk and triangulate that candidate a second time (same walks, same DLT, same solve, same failure): it starts at the 3-subset fallback on the list the round already holds and goes on with the later starting observations (0: the whole sequential N-view step, rounds 3-5) */
#endif
#ifndef EG3D_PAR_CANDIDATES
#define EG3D_PAR_CANDIDATES 1 /* look-ahead rounds: when a step's first starting observation dies, the others are walked several at a time (stepn_walks_par) instead of one pass of the wave each */
#endif
#ifndef EG3D_WINDOW_ROUNDS
#define EG3D_WINDOW_ROUNDS 0 /* batches of ADD solves: a window of requests ends where a round of the solver ends (add_solves). Measured (round 6, profiles/r06_experiments/ab7_par_candidates_window_rounds.txt): bit-exact and SLOWER, C3' 38.6-38.8 -> 39.2-39.4 ms (the extra windows cost more than the rounds saved): off, kept as an option */
#endif
#ifndef EG3D_LOOKAHEAD
#define EG3D_LOOKAHEAD 8 /* steps walked ahead per round (<= 8, and <= 64 / observations of the end point) */
#endif
#ifndef EG3D_LA_RESUME
#define EG3D_LA_RESUME 8 /* look-ahead depth after a redone round (0 = off for the rest of the following call: rounds 3-5). Round 6, light timing build: the sequential N-view steps were 17 % of the chain clocks for 215 k steps against 21 % for the 1.2 M steps of the look-ahead rounds (a step on its own pays a whole DLT stream and a solver batch); A/B 0 / 2 / 4 / 8: C3' 43.7 / 42.5 / 42.3 / 42.4 ms, C2 6.22 / 5.50 / 5.42 / 5.15 ms */
#endif
#ifndef EG3D_SIDE_WALK_BATCH
#define EG3D_SIDE_WALK_BATCH 0 /* side walks: the whole-segment tests of four consecutive walk steps in one pass over the lanes. Measured (round 5, profiles/r05_experiments/k3b_variants.txt): bit-exact, and SLOWER - C3' 44.7 against 43.4-43.5 ms: the sequential half of a step (partial segment, selection) is what a step costs, and most hits are on the partial segment; kept as a measured option */
#endif
#ifndef EG3D_PAR_APPEND
#define EG3D_PAR_APPEND 1 /* a followed point's observations are stored by m lanes at once instead of m stores by every lane (C3' 43.7-43.9 -> 43.4-43.5 ms) */
#endif
#ifndef EG3D_SPEC_FOLLOW
#define EG3D_SPEC_FOLLOW 1 /* chain following: walk up to 4 steps ahead, then triangulate them together */
#endif
// GN_KEEP: see gn_round (0 = standard build; 4 = the wide build keeps the rows of up to four chunks in registers)
// SCENE = the class of scenes an instantiation serves (the host picks it per context, launch_k3b):
//   0  small: <= 28 views and polylines of <= 512 vertices — the solver's long-request path (a point has at most one
//      observation per view, so no solve exceeds a packed round) and the side walks over polylines that do not fit
//      the LDS staging area are compiled out;
//   1  general: everything;
//   2  many views (>= 29) and polylines of <= 512 vertices: the N-view step's lists never fit LDS there (2 V + 8 > 64
//      observations), so chain following is always one step at a time — the look-ahead rounds are compiled out —,
//      the speculative central solves are always windowed, and the unstaged side walks are compiled out as in 0.
// What a scene cannot execute is not free in a 45-70 k-instruction kernel: register allocation and the instruction
// cache both see it (C3': 50.4 -> 47.5 ms with SCENE 0; C4: 1828 -> 1732 ms per step in flight with SCENE 2).
template <int GN_KEEP, int SCENE>
struct TeamWaveT {
  static constexpr bool LONG_GN = SCENE != 0;
  static constexpr int kPreIt = (SCENE == 2 || EG3D_GN_PRECHECK_ALL) ? EG3D_GN_PRECHECK_IT : 30;  // eg3d_dev_coopgn.h
  static constexpr bool kSlotStep = EG3D_WAVE_SLOT_STEP != 0;
  static constexpr bool kSpecFollow = EG3D_SPEC_FOLLOW != 0 && SCENE != 2;
  CoopLds* L;
  __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63u); }
  __device__ __forceinline__ int size() const { return 64; }
  __device__ __forceinline__ void sync() const { __syncthreads(); }
  __device__ __forceinline__ void bind(Chain& c) const {
    if constexpr (SCENE != 2)
      if (c.tmp_cap <= EG3D_COOP_ROWS) c.tmp_a = L->tmp_a;
  }
  __device__ __forceinline__ bool lazy_presolve(const DevScene& s) const {
    if constexpr (SCENE == 2) return true;
    return eg3d::lazy_presolve(s);
  }
  __device__ __forceinline__ int rank(bool flag, int& total) const {
    const unsigned long long m = __ballot(flag);
    total = __popcll(m);
    return __popcll(m & ((1ull << lane()) - 1ull));
  }
  template <class T>
  __device__ __forceinline__ T uni(const T& v) const {
    static_assert(sizeof(T) % 4 == 0, "uni(): whole dwords");
    union {
      T t;
      int w[sizeof(T) / 4];
    } u;
    u.t = v;
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) u.w[i] = __builtin_amdgcn_readfirstlane(u.w[i]);
    return u.t;
  }
  // uniform section: every lane runs the same decomposition on the same LDS words (slot 0)
  __device__ __forceinline__ void dlt(const float* P1, float x1, float y1, const float* P2, float x2, float y2,
                                      double X0[3]) const {
#if EG3D_DLT_HOT_IN_LDS
    dlt_rare(P1, x1, y1, P2, x2, y2, X0);
#else
    dlt2(P1, x1, y1, P2, x2, y2, X0);
#endif
  }
  __device__ __forceinline__ void dlt_rare(const float* P1, float x1, float y1, const float* P2, float x2, float y2,
                                           double X0[3]) const {
    // ONE lane runs the decomposition on slot 0 (64 lanes writing the same LDS words would serialise); the start
    // point is then broadcast
    __syncthreads();
    double r[3] = {0, 0, 0};
#if EG3D_DLT_GRP
    dlt2_grp8(L->dltg, lane() < 8, P1, x1, y1, P2, x2, y2, r);  // the first group of 8 lanes
#else
    typedef __attribute__((address_space(3))) double* lds_dp;
    if (lane() == 0) dlt2_mem(P1, x1, y1, P2, x2, y2, (lds_dp)&L->dlt_work[0][0], r);
#endif
    X0[0] = lane_bcast(r[0], 0);
    X0[1] = lane_bcast(r[1], 0);
    X0[2] = lane_bcast(r[2], 0);
    __syncthreads();
  }
  // OR over the lanes of a small flag word (the EG3D_FLAG_* bits 0..4 the expand stage raises): one ballot per bit
  __device__ __forceinline__ uint32_t or_reduce(uint32_t v) const {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t b = 1; b <= 16u; b <<= 1)
      if (__ballot((v & b) != 0)) r |= b;
    return r;
  }
  template <class Pred>
  __device__ __forceinline__ int leading_true(int m, Pred pred) const {
    int cnt = 0;
    for (int j0 = 0; j0 < m; j0 += 64) {
      const int j = j0 + lane();
      const unsigned long long mask = __ballot(j < m && pred(j));
      const unsigned long long inv = ~mask;
      const int lead = inv ? (__ffsll((long long)inv) - 1) : 64;
      cnt += lead;
      if (lead < 64) break;
    }
    return cnt;
  }
  __device__ __forceinline__ uint64_t valid_mask(const Chain& c, int base, int end) const {
    const int k = base + lane();
    return __ballot(k < end && c.cand[k].valid != 0);
  }
  __device__ __forceinline__ int group_size(int n_items) const {
    int g = 1;
    while (g < 16 && g * 2 * n_items <= 64) g <<= 1;
    return g;
  }
  __device__ __forceinline__ void group_best(int G, float& d, PlPt& p) const {
    for (int o = 1; o < G; o <<= 1) {
      const float od = __shfl_xor(d, o);
      const uint32_t os = (uint32_t)__shfl_xor((int)p.seg, o);
      const float ox = __shfl_xor(p.x, o), oy = __shfl_xor(p.y, o);
      if (od < d || (od == d && os < p.seg)) {
        d = od;
        p.seg = os;
        p.x = ox;
        p.y = oy;
      }
    }
  }
  __device__ __forceinline__ uint32_t excl_scan(uint32_t v, uint32_t& total) const {
    const uint32_t pre = (uint32_t)wave_incl_scan((int)v);
    total = lane_bcast(pre, 63);
    return pre - v;
  }
  // Staging for the side walks of ONE attachment (all of them walk the same polyline of the same view): the polyline's
  // vertices (when they fit) and the epipolar lines of the chain points on either side of ci — the lower side
  // (ci-1, ci-2, ... >= lo) in the first half of the staging area, the upper side (ci+1, ... < hi) in the second —
  // copied to LDS once by all lanes. (Round 3 staged per side walk: four times per attachment.)
  static constexpr int kEpiHalf = EG3D_STAGE_EPI / 2;
  __device__ __forceinline__ void walk_stage(const DevScene& s, Chain& c, int view, const PlRef& pl, int lo, int ci,
                                             int hi) const {
    __syncthreads();
    if (pl.n <= EG3D_STAGE_VTX)  // (always, in the small-scene build; the test keeps a stray long polyline from overrunning LDS)
      for (uint32_t i = (uint32_t)lane(); i < pl.n; i += 64) L->walk.vtx[i] = pl.v[i];
    const int n_lo = ci - lo < kEpiHalf ? (ci - lo < 0 ? 0 : ci - lo) : kEpiHalf;
    const int n_hi = hi - ci - 1 < kEpiHalf ? (hi - ci - 1 < 0 ? 0 : hi - ci - 1) : kEpiHalf;
    for (int t = lane(); t < n_lo + n_hi; t += 64) {
      const bool low = t < n_lo;
      const int pt = low ? ci - 1 - t : ci + 1 + (t - n_lo);
      const int slot = low ? t : kEpiHalf + (t - n_lo);
      const ViewCand& ve = c.cand[c.head + pt];
      L->walk.epi[slot][0] = ve.eok ? 1.0f : 0.0f;
      L->walk.epi[slot][1] = ve.ea;
      L->walk.epi[slot][2] = ve.eb;
      L->walk.epi[slot][3] = ve.ec;
    }
    __syncthreads();
  }
  // One side walk from the staged data, walked through address_space(3) pointers (ds_read).
  __device__ __forceinline__ int side_walk(const DevScene& s, Chain& c, int view, const PlRef& pl, const Obs& from,
                                           uint32_t direction, int lo, int ci, int hi, bool towards_start, Pending* out) const {
    typedef const __attribute__((address_space(3))) float* lds_fp;
    typedef const __attribute__((address_space(3))) f2* lds_f2p;
    const int count = towards_start ? ci - lo : hi - ci - 1;
    const bool fits = SCENE != 1 || pl.n <= EG3D_STAGE_VTX;  // scene classes 0 and 2: every polyline fits (host-checked)
    const int staged = count < kEpiHalf ? (count < 0 ? 0 : count) : kEpiHalf;
    const lds_fp epi = (lds_fp)&L->walk.epi[towards_start ? 0 : kEpiHalf][0];
    // next hit of the line towards `direction`, SEGMENT-PARALLEL: lane 0 tests the partial segment
    // from the current position, lane k the k-th whole segment beyond it; the first lane (walking
    // order) whose test reports a hit or a quasi-parallel stop decides — exactly the sequential
    // scan of walk_by_line, one test deep instead of one per segment
    auto walk = [](const auto& p, const PlPt& from, uint32_t dir, float la, float lb, float lc, PlPt& nx) -> uint32_t {
      const bool to_start = dir == p.start;
      if (!to_start && dir != p.end) return WALK_BAD_DIR;  // Q15
      const LineDir ld = line_dir(la, lb);
      const uint32_t lane = threadIdx.x & 63u;
      // candidates in walking order: towards start: 1 + from.seg ; towards end: 1 + (n - 2 - from.seg)
      const uint32_t total = 1u + (to_start ? from.seg : (p.n - 2u - from.seg));
      for (uint32_t k0 = 0; k0 < total; k0 += 64) {
        const uint32_t k = k0 + lane;
        uint32_t r = 0, seg = 0;
        float hx = 0.f, hy = 0.f;
        if (k < total) {
          float x1, y1, x2, y2;
          if (k == 0) {
            x1 = from.x;
            y1 = from.y;
            const uint32_t vi = to_start ? from.seg : from.seg + 1u;
            x2 = p.v[vi].x;
            y2 = p.v[vi].y;
            seg = from.seg;
          } else if (to_start) {
            const uint32_t i = from.seg - (k - 1u);  // segment (v[i], v[i-1]), i >= 1
            x1 = p.v[i].x;
            y1 = p.v[i].y;
            x2 = p.v[i - 1].x;
            y2 = p.v[i - 1].y;
            seg = i - 1u;
          } else {
            const uint32_t i = from.seg + k;  // segment (v[i], v[i+1]), i <= n-2
            x1 = p.v[i].x;
            y1 = p.v[i].y;
            x2 = p.v[i + 1].x;
            y2 = p.v[i + 1].y;
            seg = i;
          }
          r = seg_line_hit_guarded(x1, y1, x2, y2, la, lb, lc, ld, hx, hy);
        }
        const unsigned long long any = __ballot(r != 0);
        if (any) {
          const int f = __ffsll((long long)any) - 1;
          const uint32_t rf = lane_bcast(r, f);
          if (rf & 2u) return WALK_QUASIPARALLEL;
          nx.seg = lane_bcast(seg, f);
          nx.x = lane_bcast(hx, f);
          nx.y = lane_bcast(hy, f);
          return WALK_FOUND;
        }
      }
      return WALK_EXTREME;
    };
    if (fits) {
      PlRefT<lds_f2p> pls;
      pls.v = (lds_f2p)&L->walk.vtx[0];
      pls.n = pl.n;
      pls.start = pl.start;
      pls.end = pl.end;
#if EG3D_SIDE_WALK_BATCH
      // FOUR walk steps per pass over the lanes. A step = the next hit of chain point i's epipolar line from the
      // current position: first the PARTIAL segment from the position to the next vertex (depends on the previous
      // step's hit), then the WHOLE segments beyond it (which do not). Quarter q of the wave (16 lanes) tests line
      // t + q against the 16 whole segments beyond the batch's starting position — all four lines at once, before any
      // of the four hits is known; the steps are then resolved in order: partial segment (uniform), else the first lane
      // of the step's quarter, among the segments still ahead of the position, whose test reported a hit or a
      // quasi-parallel stop. Same tests on the same operands in the same order of precedence as one step at a time
      // (the whole-segment tests of a line do not depend on where its walk starts); a step whose 16-segment window is
      // exhausted on a longer polyline takes the one-step walk from the current position.
      {
        const bool to_start = direction == pls.start;
        if (to_start || direction == pls.end) {
          const int lane_i = lane(), q = lane_i >> 4, sl = lane_i & 15;
          int cnt = 0, t = 0;
          PlPt pos;
          pos.seg = from.seg;
          pos.x = from.x;
          pos.y = from.y;
          bool ended = false;
          while (!ended && t < staged) {
            const int nb = staged - t < 4 ? staged - t : 4;
            const uint32_t seg0 = pos.seg;
            // ---- the whole segments beyond seg0 against the lines of this batch
            uint32_t r = 0, wseg = 0;
            float hx = 0.f, hy = 0.f;
            if (q < nb) {
              const float la = epi[4 * (t + q) + 1], lb = epi[4 * (t + q) + 2], lc = epi[4 * (t + q) + 3];
              bool valid;
              uint32_t i;
              if (to_start) {
                valid = seg0 >= (uint32_t)sl + 1u;  // segment (v[i], v[i-1]), i = seg0 - sl >= 1
                i = seg0 - (uint32_t)sl;
                wseg = i - 1u;
              } else {
                i = seg0 + 1u + (uint32_t)sl;       // segment (v[i], v[i+1]), i <= n - 2
                valid = i + 1u < pls.n;
                wseg = i;
              }
              if (valid) {
                const uint32_t i1 = to_start ? i - 1u : i + 1u;
                r = seg_line_hit_guarded(pls.v[i].x, pls.v[i].y, pls.v[i1].x, pls.v[i1].y, la, lb, lc, line_dir(la, lb), hx, hy);
              }
            }
            const unsigned long long any = __ballot(r != 0);
            const uint32_t beyond0 = to_start ? seg0 : (pls.n - 2u - seg0);  // whole segments beyond the starting position
            // ---- the steps of the batch, in order
            int qq = 0;
            for (; qq < nb; qq++) {
              if (epi[4 * (t + qq)] == 0.0f) {
                ended = true;
                break;
              }
              const float la = epi[4 * (t + qq) + 1], lb = epi[4 * (t + qq) + 2], lc = epi[4 * (t + qq) + 3];
              const uint32_t vi = to_start ? pos.seg : pos.seg + 1u;
              float px = 0.f, py = 0.f;
              const uint32_t rp = seg_line_hit_guarded(pos.x, pos.y, pls.v[vi].x, pls.v[vi].y, la, lb, lc, line_dir(la, lb), px, py);
              if (rp & 2u) {
                ended = true;
                break;
              }
              PlPt nx;
              if (rp & 1u) {
                nx.seg = pos.seg;
                nx.x = px;
                nx.y = py;
              } else {
                const uint32_t adv = to_start ? seg0 - pos.seg : pos.seg - seg0;  // whole segments of the window already behind
                const uint32_t m = adv < 16u ? ((uint32_t)(any >> (16 * qq)) & 0xffffu & (0xffffu << adv)) : 0u;
                if (m) {
                  const int f = 16 * qq + __ffs((int)m) - 1;
                  const uint32_t rf = lane_bcast(r, f);
                  if (rf & 2u) {
                    ended = true;
                    break;
                  }
                  nx.seg = lane_bcast(wseg, f);
                  nx.x = lane_bcast(hx, f);
                  nx.y = lane_bcast(hy, f);
                } else if (beyond0 <= 16u) {
                  ended = true;  // every segment up to the extreme was tested: the walk ends there
                  break;
                } else {
                  // the window is exhausted on a long polyline: one step the plain way, then a new batch from there
                  const uint32_t w = walk(pls, pos, direction, la, lb, lc, nx);
                  if (!(w & WALK_FOUND)) {
                    ended = true;
                    break;
                  }
                  Pending& pd = out[cnt++];
                  pd.o.view = view;
                  pd.o.pl = from.pl;
                  pd.o.seg = nx.seg;
                  pd.o.x = nx.x;
                  pd.o.y = nx.y;
                  pd.ok = 0;
                  pos = nx;
                  qq++;
                  break;
                }
              }
              Pending& pd = out[cnt++];
              pd.o.view = view;
              pd.o.pl = from.pl;
              pd.o.seg = nx.seg;
              pd.o.x = nx.x;
              pd.o.y = nx.y;
              pd.ok = 0;
              pos = nx;
            }
            t += qq;
          }
          if (ended || count <= staged) return cnt;
          // more chain points than the staging area holds lines for (never on the bench's scenes): the rest one step at a
          // time from the candidate array, continuing from the current position
          {
            Obs cur = from;
            cur.seg = pos.seg;
            cur.x = pos.x;
            cur.y = pos.y;
            const int ci2 = towards_start ? ci - staged : ci + staged;
            return cnt + walk_side_candidates_core(s, c, pls, epi, 0, view, cur, direction, lo, ci2, hi, towards_start, out + cnt, walk);
          }
        }
      }
#endif
      return walk_side_candidates_core(s, c, pls, epi, staged, view, from, direction, lo, ci, hi, towards_start, out,
                                       walk);
    }
    if constexpr (SCENE != 1) return 0;  // unreachable in those builds (fits is a constant there)
    return walk_side_candidates_core(s, c, pl, epi, staged, view, from, direction, lo, ci, hi, towards_start, out, walk);
  }
  // The walk phase of the starting observations st0 .. n-1 of an N-view step, SEVERAL CANDIDATES AT ONCE (round 6). The
  // sequential order tries them one after the other and takes the first that keeps >= 3 observations; each try is a
  // pass of the wave in which n - 1 lanes walk — and a following ENDS with a step in which every candidate is tried
  // and dies (4 % of the chain clocks on C3'). Here lane (g, i) of a pass is observation i of candidate st = base + g,
  // 64 / n candidates per pass: every lane of a group advances the group's starting observation itself (the same
  // uniform walk, redundantly), then follows its own observation; the first group in order whose walks keep >= 3
  // wins, its survivors are compacted in observation order exactly as stepn_walks does, and only the diagnostic flags
  // of the candidates up to the winner count (the sequential order never ran the later ones). Returns m (0: all dead).
  __device__ __forceinline__ int stepn_walks_par(const DevScene& s, const Obs* co_all, int n, int st0, const uint32_t* dirs,
                                                 Obs* sel, uint32_t& flags, int& st_used) const {
    const int per = 64 / n;
    if (per < 2) {
      int m = 0;
      for (int st = st0; st < n && m == 0; st++) {
        m = stepn_walks(*this, s, co_all, n, st, dirs, sel, n, flags);
        st_used = st;
      }
      return m;
    }
    const int g = lane() / n, i = lane() - g * n;
    for (int base = st0; base < n; base += per) {
      const int st = base + g;
      const bool valid = g < per && st < n;
      bool dead = true, bad = false, found = false, bad_i = false;
      PlPt q;
      q.seg = 0;
      q.x = q.y = 0.f;
      Obs so, r;
      so.view = 0;
      so.pl = so.seg = 0;
      so.x = so.y = 0.f;
      r = so;
      if (valid) {
        so = co_all[st];
        const PlRef ps = polyline_of(s, so.view, so.pl);
        PlPt p;
        p.seg = so.seg;
        p.x = so.x;
        p.y = so.y;
        const uint32_t w = walk_by_distance(ps, p, dirs[so.view], EG3D_FOLLOW_STEP, q);
        bad = (w & WALK_BAD_DIR) != 0;
        dead = (w & WALK_EXTREME) != 0;
        if (!dead && i != st) {
          const Obs co = co_all[i];
          float la, lb, lc;
          if (epiline(s.F, s.F_valid, s.n_views, so.view, co.view, q.x, q.y, la, lb, lc)) {
            const PlRef pk = polyline_of(s, co.view, co.pl);
            PlPt cp, rp;
            cp.seg = co.seg;
            cp.x = co.x;
            cp.y = co.y;
            const uint32_t wr = walk_by_line(pk, cp, dirs[co.view], la, lb, lc, true, EG3D_FOLLOW_MIN, EG3D_FOLLOW_MAX, rp);
            bad_i = (wr & WALK_BAD_DIR) != 0;
            if (wr & WALK_FOUND) {
              found = true;
              r.view = co.view;
              r.pl = co.pl;
              r.seg = rp.seg;
              r.x = rp.x;
              r.y = rp.y;
            }
          }
        }
      }
      const unsigned long long fm = __ballot(found);
      const unsigned long long gm = ((1ull << n) - 1ull) << (valid ? g * n : 0);  // this lane's group (n <= 32 here)
      const int cnt = valid ? __popcll(fm & gm) : 0;
      const bool alive = valid && !dead && 1 + cnt >= 3;
      const unsigned long long am = __ballot(alive);
      const unsigned long long bm = __ballot(valid && (bad_i || bad));
      const int gw = am ? (__ffsll((long long)am) - 1) / n : per;  // the winning group (per: none in this pass)
      // flags of the candidates the sequential order would have run: groups <= gw
      {
        const int upto = gw < per ? (gw + 1) * n : per * n;
        const unsigned long long lanes = upto >= 64 ? ~0ull : ((1ull << upto) - 1ull);
        if (bm & lanes) flags |= 8u;
      }
      if (gw < per) {
        if (g == gw) {
          if (i == 0) {
            Obs o0;
            o0.view = so.view;
            o0.pl = so.pl;
            o0.seg = q.seg;
            o0.x = q.x;
            o0.y = q.y;
            sel[0] = o0;
          }
          if (found) sel[1 + __popcll(fm & gm & ((1ull << lane()) - 1ull))] = r;
        }
        const int m = 1 + lane_bcast(cnt, gw * n);
        st_used = base + gw;
        __syncthreads();
        return m;
      }
    }
    __syncthreads();
    return 0;
  }
  // append a followed point at the chain's front / back (the checks of follow_front / follow_back)
  __device__ __forceinline__ bool follow_append(Chain& c, bool front, const Obs* list, int m, const float X[3]) const {
    if (front ? (c.head <= 0) : (c.head + c.len >= c.cap_pts)) {
      c.flags |= 1u;
      return false;
    }
    ChainPt np;
#if EG3D_PAR_APPEND
    // the new point's block is reserved for m + 1 observations (point_reserve: the smallest power of two >= 4 that holds
    // them), so the m appends of new_point_from_list never relocate: lane i stores observation i (the sequential form has
    // every lane store all m, one after the other)
    np.X[0] = X[0];
    np.X[1] = X[1];
    np.X[2] = X[2];
    point_init(np);
    if (!point_reserve(c, np, (uint32_t)m + 1)) return false;
    for (int i = lane(); i < m; i += 64) c.pool[np.off + (uint32_t)i] = list[i];
    np.nobs = (uint32_t)m;
#else
    if (!new_point_from_list(c, np, list, m, X)) return false;
#endif
    if (front) {
      c.head--;
      c.pts[c.head] = np;
    } else {
      c.pts[c.head + c.len] = np;
    }
    c.len++;
#if EG3D_PAR_APPEND
    __syncthreads();  // the observations were stored by different lanes: visible to all before the next step reads them
#endif
    return true;
  }
  // Chain following (follow_direction_vector_start/_end, plg_matching.cpp:771-795) with LOOK-AHEAD.
  // The walks of step t+1 start from the observations step t FOUND, not from its triangulated X,
  // so up to D = 8 steps are walked ahead first (each: the first starting observation whose walks
  // keep >= 3 observations — exactly the candidate the sequential N-view step triangulates first);
  // their D initial DLTs then run side by side on D lanes (one DLT's worth of instructions instead
  // of D) and their D all-observation Gauss-Newton solves as D groups of one cooperative batch.
  // Steps are accepted in order while their triangulation succeeds; the first one that fails is
  // redone by the sequential N-view step (3-subset fallback, later candidates), and the steps
  // walked beyond it are dropped (their diagnostic flags too). Measured: a following call adds
  // 3.4-4.7 points and >95 % of the triangulations succeed.
  __device__ __forceinline__ int follow(const DevScene& s, Chain& c, bool front) const {
    const uint32_t* dirs = front ? c.start_dirs : c.end_dirs;
    int added = 0;
    // look-ahead depth limit: EG3D_LOOKAHEAD to start with; after a round whose step had to be redone
    // the sequential step runs once, then look-ahead resumes at depth EG3D_LA_RESUME (0 = stays off for
    // the rest of the call) and doubles with every round that is accepted whole
    int d_cap = EG3D_LOOKAHEAD;
    bool look_ahead = true;
    for (;;) {
      const ChainPt& endpt = front ? chain_at(c, 0) : chain_at(c, c.len - 1);
      const int n_end = (int)endpt.nobs;
      int D = n_end > 0 ? EG3D_COOP_ROWS / n_end : 0;
      if (D > d_cap) D = d_cap;
      bool seq = !look_ahead || D < 2 || c.tmp_a != L->tmp_a;  // long observation lists (or lists not in LDS): plain steps
      if (!seq) {
      // ---- stage 1: walk ahead (lists of step j at tmp_a + j * n_end; a step keeps <= n_end obs)
      int Deff = 0;
      uint32_t fl_dead = 0;  // walk flags of the step that died (merged when the following ends there)
      const uint64_t tq0 = EG3D_TICK();
      {
        const Obs* prev = c.pool + endpt.off;
        int nprev = n_end;
        for (int j = 0; j < D; j++) {
          Obs* sel = L->tmp_a + j * n_end;
          uint32_t fl = 0;
          int m = 0;
#ifdef EG3D_ONE_SECTION
          const uint64_t tdd0 = EG3D_TICK();  // light timing build: section 11 = the walks of the step that DIES (every starting observation tried)
#endif
          int st_used = 0;
#if EG3D_PAR_CANDIDATES
          m = stepn_walks(*this, s, prev, nprev, 0, dirs, sel, n_end, fl);  // (nearly every step that lives, lives on its first candidate)
          if (m == 0 && nprev > 1) m = stepn_walks_par(s, prev, nprev, 1, dirs, sel, fl, st_used);
#else
          for (int st = 0; st < nprev && m == 0; st++) {
            m = stepn_walks(*this, s, prev, nprev, st, dirs, sel, n_end, fl);
            st_used = st;
          }
#endif
          if (m == 0) {
#ifdef EG3D_ONE_SECTION
            EG3D_SEC_ADD(c.tsec, 11, EG3D_TICK() - tdd0);
#endif
            fl_dead = fl;
            break;
          }
          if (lane() == 0) {
            L->la_m[j] = m;
            L->la_fl[j] = fl;
            L->la_st[j] = st_used;
          }
          Deff++;
          prev = sel;
          nprev = m;
        }
      }
      if (Deff == 0) {  // no starting observation survives its walks: the following ends here
        c.flags |= fl_dead;
        return added;
      }
      __syncthreads();  // la_m / la_fl
      // ---- stage 2: the Deff initial DLTs, list j on lane j
      const uint64_t tq1 = EG3D_TICK();
      EG3D_SEC_ADD(c.tsec, 1, tq1 - tq0);
      double X0[3] = {0, 0, 0};
      uint32_t dfl = 0;
#if EG3D_DLT_GRP
      // list j on the 8 lanes of group j (every lane of the group selects the two observations; the decomposition is
      // spread over the group: dlt2_grp8); the start point and the flag then move to lane j, where request j lives
      {
        const int gj = lane() >> 3;
        const bool on = gj < Deff;
        const float* P1 = s.cam_P;
        const float* P2 = s.cam_P;
        float gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f;
        if (on) {
          const Obs* a = L->tmp_a + gj * n_end;
          const int n = L->la_m[gj];
          int mi = 0;
          int32_t mv = a[0].view;
          for (int i = 0; i < n; i++)
            if (a[i].view < mv) {
              mv = a[i].view;
              mi = i;
            }
          const int la = n - 1;
          if (a[mi].view == a[la].view) dfl = 16u;
          P1 = s.cam_P + (size_t)a[mi].view * 16;
          gx1 = a[mi].x;
          gy1 = a[mi].y;
          P2 = s.cam_P + (size_t)a[la].view * 16;
          gx2 = a[la].x;
          gy2 = a[la].y;
        }
        double Xg[3] = {0, 0, 0};
        dlt2_grp8(L->dltg, on, P1, gx1, gy1, P2, gx2, gy2, Xg);
        const int src = (lane() & 7) * 8;  // lane j < 8 takes group j's result
        X0[0] = (double)__shfl((float)Xg[0], src);  // DLT results are float-valued
        X0[1] = (double)__shfl((float)Xg[1], src);
        X0[2] = (double)__shfl((float)Xg[2], src);
        dfl = (uint32_t)__shfl((int)dfl, src);
        if (lane() >= Deff) dfl = 0;
      }
#else
      if (lane() < Deff) {
        const Obs* a = L->tmp_a + lane() * n_end;
        const int n = L->la_m[lane()];
        int mi = 0;
        int32_t mv = a[0].view;
        for (int i = 0; i < n; i++)
          if (a[i].view < mv) {
            mv = a[i].view;
            mi = i;
          }
        const int la = n - 1;
        if (a[mi].view == a[la].view) dfl = 16u;
#if EG3D_DLT_HOT_IN_LDS
        typedef __attribute__((address_space(3))) double* lds_dp;
        dlt2_mem(s.cam_P + (size_t)a[mi].view * 16, a[mi].x, a[mi].y, s.cam_P + (size_t)a[la].view * 16, a[la].x, a[la].y,
                 (lds_dp)&L->dlt_work[lane() & 7][0], X0);
#else
        dlt2(s.cam_P + (size_t)a[mi].view * 16, a[mi].x, a[mi].y, s.cam_P + (size_t)a[la].view * 16, a[la].x, a[la].y, X0);
#endif
      }
#endif  // EG3D_DLT_GRP
      // ---- stage 3: the Deff Gauss-Newton solves as one batch (request j on lane j)
      const uint64_t tq2 = EG3D_TICK();
      EG3D_SEC_ADD(c.tsec, 5, tq2 - tq1);
      {
        const bool want = lane() < Deff;
        const float X0f[3] = {(float)X0[0], (float)X0[1], (float)X0[2]};  // DLT results are float-valued
        float Xr[3];
        const bool ok = coop_gn_groups<GN_KEEP, LONG_GN, kPreIt>(s.cam_P, *L, want, L->tmp_a + (want ? lane() : 0) * n_end,
                                       want ? L->la_m[lane()] : 0, false, 0, 0.f, 0.f, X0f, Xr);
        // results stay in the request table: L->res_ok[j], L->x0[j]
        (void)ok;
        (void)Xr;
      }
      // ---- stage 4: accept in order
      EG3D_SEC_ADD(c.tsec, 6, EG3D_TICK() - tq2);
      bool redo = false, stop = false;
      int redo_j = 0;
      for (int j = 0; j < Deff; j++) {
        const uint32_t dflj = lane_bcast(dfl, j);
        if (L->res_ok[j]) {
          const float X[3] = {L->x0[j][0], L->x0[j][1], L->x0[j][2]};
          c.flags |= L->la_fl[j] | dflj;
          if (!follow_append(c, front, L->tmp_a + j * n_end, L->la_m[j], X)) {
            stop = true;
            break;
          }
          added++;
#if defined(EG3D_SECTION_TIMING) && !defined(EG3D_ONE_SECTION)
          EG3D_SEC_ADD(c.tsec, 11, 1ull << 16);  // diagnostic: steps accepted from a look-ahead round
#endif
        } else {
#if defined(EG3D_SECTION_TIMING) && !defined(EG3D_ONE_SECTION)
          EG3D_SEC_ADD(c.tsec, 11, 1ull << 32);  // diagnostic: look-ahead rounds that ended in a redo
#endif
          redo = true;  // sequential N-view step from the chain's current end (same walks, then the
          redo_j = j;   // 3-subset fallback and the later candidates)
          break;
        }
      }
#if EG3D_REDO_SKIP
      // The step that failed is candidate la_st[j] of the sequential N-view step from the chain's current end (the
      // steps before it were appended, so that end IS the list it was walked from): its walks, its DLT and its
      // all-observation solve would be repeated with the same operands and fail the same way. Go on where the
      // sequential order goes on after that failure: the 3-subset fallback on the list, then the later candidates.
      if (redo && !stop) {
        const int mj = L->la_m[redo_j], stj = L->la_st[redo_j];
        const uint32_t flj = L->la_fl[redo_j] | lane_bcast(dfl, redo_j);
        Obs keep;
        keep.view = 0;
        keep.pl = keep.seg = 0;
        keep.x = keep.y = 0.f;
        if (lane() < mj) keep = L->tmp_a[redo_j * n_end + lane()];
        __syncthreads();
        if (lane() < mj) L->tmp_a[lane()] = keep;
        __syncthreads();
        c.flags |= flj;
        const ChainPt& e2 = front ? chain_at(c, 0) : chain_at(c, c.len - 1);
        float X[3];
        const uint64_t tsq0 = EG3D_TICK();
        int m = stepn_fallback(*this, s, c.tmp_a, mj, c.tmp_b, c.tmp_mask, X, c.flags);
        if (m == 0) m = stepn_chain(*this, s, c, e2, dirs, X, stj + 1);
        EG3D_SEC_ADD(c.tsec, 15, EG3D_TICK() - tsq0);
        if (m == 0) return added;
        if (!follow_append(c, front, c.tmp_a, m, X)) return added;
        added++;
        look_ahead = EG3D_LA_RESUME >= 2;
        d_cap = EG3D_LA_RESUME;
        continue;
      }
#endif
      __syncthreads();  // the lists / results are rewritten next
      if (stop) return added;
      if (!redo) {
        if (Deff < D) {  // step Deff died in its walks after Deff accepted steps: the following ends
          c.flags |= fl_dead;
          return added;
        }
        if (d_cap < EG3D_LOOKAHEAD) d_cap *= 2;
        continue;
      }
      seq = true;  // redo the failed step with the sequential N-view step; where the first candidate's
      look_ahead = EG3D_LA_RESUME >= 2;  // (0: look-ahead stays off for the rest of this following call)
      d_cap = EG3D_LA_RESUME;
      }
      if (seq) {
        const ChainPt& e2 = front ? chain_at(c, 0) : chain_at(c, c.len - 1);
        float X[3];
        const uint64_t tsq0 = EG3D_TICK();
        const int m = stepn_chain(*this, s, c, e2, dirs, X);
        EG3D_SEC_ADD(c.tsec, 15, EG3D_TICK() - tsq0);
#if defined(EG3D_SECTION_TIMING) && !defined(EG3D_ONE_SECTION)
        EG3D_SEC_ADD(c.tsec, 11, 1ull);  // diagnostic: sequential N-view steps
#endif
        if (m == 0) return added;
        if (!follow_append(c, front, c.tmp_a, m, X)) return added;
        added++;
      }
    }
  }
  // uniform section: all lanes hold the same (a, n, X0) and receive the same answer
  __device__ __forceinline__ bool gn_array(const DevScene& s, const Obs* a, int n, const double X0[3],
                                           float Xout[3]) const {
    // one request (lane 0), the whole wave on its rows
    const float X0f[3] = {(float)X0[0], (float)X0[1], (float)X0[2]};  // callers pass float-valued starts
    float Xr[3];
    const bool ok = coop_gn_groups<GN_KEEP, LONG_GN, kPreIt>(s.cam_P, *L, lane() == 0, a, n, false, 0, 0.f, 0.f, X0f, Xr);
    Xout[0] = lane_bcast(Xr[0], 0);
    Xout[1] = lane_bcast(Xr[1], 0);
    Xout[2] = lane_bcast(Xr[2], 0);
    return lane_bcast(ok ? 1 : 0, 0) != 0;
  }
  __device__ __forceinline__ bool add_array(const DevScene& s, const Obs* a, int n, const Obs& extra, const float X0[3],
                                            float Xout[3]) const {
    float Xr[3];
    const bool ok = coop_gn_groups<GN_KEEP, LONG_GN, kPreIt>(s.cam_P, *L, lane() == 0, a, n, true, extra.view, extra.x, extra.y, X0, Xr);
    Xout[0] = lane_bcast(Xr[0], 0);
    Xout[1] = lane_bcast(Xr[1], 0);
    Xout[2] = lane_bcast(Xr[2], 0);
    return lane_bcast(ok ? 1 : 0, 0) != 0;
  }
  __device__ __forceinline__ bool add_one(const DevScene& s, const Chain& c, const ChainPt& p, const Obs& extra,
                                          float Xout[3]) const {
    const float X0f[3] = {p.X[0], p.X[1], p.X[2]};
    float Xr[3];
    const bool ok = coop_gn_groups<GN_KEEP, LONG_GN, kPreIt>(s.cam_P, *L, lane() == 0, c.pool + p.off, (int)p.nobs, true, extra.view, extra.x,
                                   extra.y, X0f, Xr);
    Xout[0] = lane_bcast(Xr[0], 0);
    Xout[1] = lane_bcast(Xr[1], 0);
    Xout[2] = lane_bcast(Xr[2], 0);
    return lane_bcast(ok ? 1 : 0, 0) != 0;
  }
  // B independent ADD solves, 64 per window, request j on lane j. A window goes cooperative
  // (rows = observations) when that needs fewer row-passes than the longest single solve;
  // otherwise each lane runs its own solve. Both produce the same bits.
  template <class Get, class Put>
  __device__ __forceinline__ void add_solves(const DevScene& s, Chain& c, int B, Get get, Put put) const {
    int take = EG3D_COOP_REQ;
    for (int w0 = 0; w0 < B; w0 += take) {
      const int j = w0 + lane();
      const ChainPt* pt = nullptr;
      Obs o;
      o.view = 0;
      o.pl = o.seg = 0;
      o.x = o.y = 0.f;
      bool want = lane() < EG3D_COOP_REQ && j < B && get(j, pt, o);
      take = EG3D_COOP_REQ;
#if EG3D_WINDOW_ROUNDS
      // More requests follow this window: end it where a ROUND of the solver ends. The solver packs whole requests into
      // rounds of <= 64 rows in lane order; a window of 32 requests usually ends in a partly filled round (32 requests of
      // 9 rows: 7 + 7 + 7 + 7 + 4), which the next window's first requests could have shared — the requests of that last
      // round are left to the next window instead (same rounds as one greedy packing of the whole batch; results do not
      // depend on the grouping).
      if (B - w0 > EG3D_COOP_REQ) {
        const int rows = want ? (int)pt->nobs + 1 : 0;
        if (!__ballot(rows > EG3D_GN_PACK_MAX)) {  // (long requests take the other path of the solver: windows as they come)
          const int pre = wave_incl_scan(rows);
          int start = 0;
          for (;;) {
            const int base = start > 0 ? lane_bcast(pre, start - 1) : 0;
            const unsigned long long fit = __ballot(lane() >= start && lane() < EG3D_COOP_REQ && pre - base <= EG3D_COOP_ROWS);
            const unsigned long long nofit = ~fit & ~((1ull << start) - 1ull) & ((1ull << EG3D_COOP_REQ) - 1ull);
            const int next = nofit ? __ffsll((long long)nofit) - 1 : EG3D_COOP_REQ;
            if (next >= EG3D_COOP_REQ || next <= start) break;  // this round reaches the window's end
            start = next;
          }
          if (start > 0) take = start;
          want = want && lane() < take;
        }
      }
#endif
      float X[3] = {0.f, 0.f, 0.f};
      float X0[3] = {0.f, 0.f, 0.f};
      if (want) {
        X0[0] = pt->X[0];
        X0[1] = pt->X[1];
        X0[2] = pt->X[2];
      }
      const bool ok = coop_gn_groups<GN_KEEP, LONG_GN, kPreIt>(s.cam_P, *L, want, want ? c.pool + pt->off : nullptr, want ? (int)pt->nobs : 0, true,
                                     o.view, o.x, o.y, X0, X);
      if (want) put(j, ok, X);
    }
  }
};
using TeamWave = TeamWaveT<0, 1>;

// Waves per SIMD the expand kernel is built for. Round 4: 4 (128 VGPRs) — CoopLds was cut to 8 LDS allocation units
// (eg3d_dev_coopgn.h) so that four single-wave workgroups really fit a SIMD: rounds 2-3 compared "2 / 3 / 4" with an LDS
// footprint that capped the residency at 3 whatever the registers, i.e. they never measured 4. At 4 the compiler
// spills 261 vector registers (352 B of scratch per lane; almost all of them around the inlined solver calls, not
// inside its loops) against 24 at 3, and the kernel moves 48.9 instead of 29.4 GB per C3' launch — and is faster on
// every workload: C3' K3b 50.3 vs 52.1 ms (47.3 vs 49.6 ms per step in flight), C2 6.85 vs 7.26 ms, the 8192-seed C4
// step 2070 vs 2233 ms, one pass over all of C4 22.7 vs 24.7 s. (2 waves with nothing spilled: 68.2 ms / 2687 ms.)
// What the extra wave hides — the dependent trips of the walks and of a solve's steps — outweighs the spill traffic:
// occupancy is the lever on this kernel. -DEG3D_K3B_WAVES=3 (tools/build_variant.sh) rebuilds the other one.
#ifndef EG3D_K3B_WAVES
#define EG3D_K3B_WAVES 4
#endif
// ---- working slices: a slot-indexed arena ---------------------------------------------------------
// A chain's working state (point headers, observation pool, candidate / pending arrays) lives in a
// SLICE of ChainLayout::total bytes. Slices belong to SLOTS, not to chains: the arena holds as many
// slices as wavefronts can be resident (a few thousand), a chain borrows one for its lifetime and the
// next chain on that slot reuses the same addresses — the arena is a few hundred MB that stays in
// L2 / Infinity Cache and in the TLB, where a slice per chain was 4 GB (C3') to tens of GB (C4) of
// first-touch traffic per launch. Slots are XCD-AFFINE: the per-XCD L2s are not coherent with each
// other, so a slice is only ever touched through ONE XCD's L2 — a wave reads its XCC id and takes a
// slot from that XCD's pool. Hand-over needs no cache maintenance then: the releasing wave waits for
// its stores to be acknowledged by that L2 (s_waitcnt vmcnt(0)) before it returns the slot, and a chain
// never reads a byte of its slice that it has not written itself (so a stale line in a CU's L1 from
// an earlier tenant is never observed). The pool of an XCD is a ring of slot ids with ticket counters:
// pop = take a ticket, then swap the cell at that position to 0 until a slot id comes out; push = take a
// ticket, then CAS the cell from 0 to the id. A pool holds at least as many slots as blocks can be
// resident on its XCD, so a pop only ever waits for a push that is already under way.
__global__ void k_pool_init(SlotPools P) {
  uint32_t* b = P.base + (size_t)blockIdx.x * P.stride;
  for (uint32_t i = threadIdx.x; i < P.ring_n; i += blockDim.x) b[32 + i] = i < P.slots_per_xcd ? i + 1u : 0u;
  if (threadIdx.x == 0) {
    b[0] = 0;
    b[16] = P.slots_per_xcd;
  }
}
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
#define EG3D_SLOT_NONE 0xffffffffu
__device__ __forceinline__ uint32_t pool_pop(const SlotPools& P, uint32_t xcc) {
  uint32_t* b = P.base + (size_t)xcc * P.stride;
  const uint32_t h = atomicAdd(&b[0], 1u);
  uint32_t* cell = &b[32 + (h & (P.ring_n - 1u))];
  for (uint32_t spin = 0; spin < (1u << 22); spin++) {  // bounded: a pool smaller than the residency is a host bug
    const uint32_t v = atomicExch(cell, 0u);
    if (v) return v - 1u;
    __builtin_amdgcn_s_sleep(16);
  }
  return EG3D_SLOT_NONE;
}
// false = the cell never emptied within the bound (cannot happen while the ring has more cells than slots; reported as
// CTR_SLOT_STARVED by the caller rather than silently losing the slot)
__device__ __forceinline__ bool pool_push(const SlotPools& P, uint32_t xcc, uint32_t slot) {
  uint32_t* b = P.base + (size_t)xcc * P.stride;
  const uint32_t t = atomicAdd(&b[16], 1u);
  uint32_t* cell = &b[32 + (t & (P.ring_n - 1u))];
  for (uint32_t spin = 0; spin < (1u << 22); spin++) {
    if (atomicCAS(cell, 0u, slot + 1u) == 0u) return true;
    __builtin_amdgcn_s_sleep(4);
  }
  return false;
}

// One wavefront per chain, launched longest-first. The finished chain is PACKED into the launch's
// staging area (point headers + its observations back to back, bump-allocated in order of completion)
// before the slot is returned: what leaves the kernel is the chain's result, 16 B per point and per
// observation written once with coalesced stores — not the working slice.
// <WAVES per SIMD the register allocation aims at, GN_KEEP>: the product instantiates <EG3D_K3B_WAVES, 0>. Round 4 measured a
// "wide" instantiation <2, 4> (256 registers: nothing spills, the Gauss-Newton rows of up to four chunks stay in registers
// between the passes of an iteration, so long solves do not recompute them): bit-exact, and SLOWER on every workload
// (C3' K3b 53.2 -> 68.2 ms, the 8192-seed C4 step 2230 -> 2687 ms): what the third wave per SIMD hides in the walks, the
// candidate search and the dependent steps of a solve outweighs the row arithmetic saved.
template <int WAVES, int GN_KEEP, int SCENE>
__global__ void __launch_bounds__(64, WAVES) k3b_expand_t(DevScene s, StageAView a, const TaskDesc* tasks,
                                                 const ChainSeed* chains, uint32_t n_chains, const uint32_t* hyp_off,
                                                 const HypResult* res, const HPoint* arena, const int32_t* map_view,
                                                 const uint32_t* map_entry, const uint32_t* map_n, ChainLayout L,
                                                 unsigned char* slices, SlotPools pools, StageBuf stage, ChainOut* outs,
                                                 uint32_t* out_points, uint32_t* out_obs, Counters* ctr,
                                                 const uint32_t* order) {
  if (blockIdx.x >= n_chains) return;
  __shared__ CoopLds lds;
  const uint32_t lane = threadIdx.x;
  const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)order[blockIdx.x]);  // longest-first schedule; results stay indexed by chain
  const uint32_t xcc = xcc_id();
  uint32_t slot = 0;
  if (lane == 0) slot = pool_pop(pools, xcc);
  slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
  ChainOut co;
  if (slot == EG3D_SLOT_NONE) {
    if (lane == 0) {
      memset(&co, 0, sizeof(co));
      outs[j] = co;
      out_points[j] = 0;
      out_obs[j] = 0;
      atomicOr(&ctr->flags, CTR_SLOT_STARVED);
    }
    return;
  }
  TeamWaveT<GN_KEEP, SCENE> tm;
  tm.L = &lds;
  if (lane == 0) {
    lds.cams_mid_range = s.cams_mid_range ? 1 : 0;
    lds.long_refused = 0;
    lds.t_start = (uint32_t)wall_clock64();
  }
  __syncthreads();
  // wave-uniform descriptors: kept in scalar registers for the chain's whole life (as vector registers they would be
  // 17 of the 168 the kernel may use, and spilled)
  const ChainSeed cs = tm.uni(chains[j]);
  const TaskDesc d = tm.uni(tasks[cs.task]);
  unsigned char* slice = slices + L.total * ((size_t)xcc * pools.slots_per_xcd + slot);
  expand_chain(tm, s, a, d, cs, tm.uni(hyp_off[cs.task]), res, arena, map_view, map_entry, map_n, L, slice, co);
  // ---- pack the result: 64 points at a time, their observations as one flat range
  unsigned long long pb = 0, ob = 0;
  if (lane == 0) {
    pb = atomicAdd(&stage.used[0], (unsigned long long)co.n_points);
    ob = atomicAdd(&stage.used[1], (unsigned long long)co.n_obs);
  }
  pb = lane_bcast(pb, 0);
  ob = lane_bcast(ob, 0);
  co.spt = pb;
  co.sobs = ob;
  if (pb + co.n_points <= stage.cap_pts && ob + co.n_obs <= stage.cap_obs) {
    __syncthreads();
    uint32_t* s_excl = (uint32_t*)&lds.prod[0][0];  // [65] first flat observation of each of the 64 points in flight
    uint32_t* s_blk = s_excl + 65;                  // [64] where each point's block starts in the pool
    const ChainPt* pts = (const ChainPt*)(slice + L.off_pts) + co.head;
    const Obs* pool = (const Obs*)(slice + L.off_pool);
    StagePt* spt = stage.pts + pb;
    Obs* sob = stage.obs + ob;
    for (uint32_t i0 = 0; i0 < co.n_points; i0 += 64) {
      const uint32_t i = i0 + lane;
      const bool act = i < co.n_points;
      ChainPt p;
      p.nobs = 0;
      p.off = 0;
      p.X[0] = p.X[1] = p.X[2] = 0.f;
      if (act) p = pts[i];
      const uint32_t incl = (uint32_t)wave_incl_scan((int)p.nobs);
      const uint32_t total = lane_bcast(incl, 63);
      s_excl[lane] = incl - p.nobs;
      s_blk[lane] = p.off;
      if (lane == 63) s_excl[64] = total;
      if (act) {
        StagePt sp;
        sp.X[0] = p.X[0];
        sp.X[1] = p.X[1];
        sp.X[2] = p.X[2];
        sp.nobs = p.nobs;
        spt[i] = sp;
      }
      __syncthreads();
      for (uint32_t f = lane; f < total; f += 64) {
        uint32_t lo = 0;  // the point whose range holds f: largest q with s_excl[q] <= f (empty points skipped)
#pragma unroll
        for (uint32_t step = 32; step; step >>= 1)
          if (s_excl[lo + step] <= f) lo += step;
        sob[f] = pool[s_blk[lo] + (f - s_excl[lo])];
      }
      __syncthreads();
      sob += total;
    }
  }
  // every store to the slice has been acknowledged by this XCD's L2 before the slot changes hands
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (lane == 0) {
    if (!pool_push(pools, xcc, slot)) atomicOr(&ctr->flags, CTR_SLOT_STARVED);
    outs[j] = co;
    out_points[j] = co.n_points;
    out_obs[j] = co.n_obs;
    if (co.flags) atomicOr(&ctr->flags, co.flags);
    if (SCENE == 0 && lds.long_refused) atomicOr(&ctr->flags, CTR_LONG_REFUSED);
    if (co.bytes) atomicAdd(&ctr->bytes, (unsigned long long)co.bytes);
    // how long this chain held its wavefront (a launch cannot be shorter than its slowest chain: reported per call)
    atomicMax(&ctr->max_chain_ticks, (uint32_t)wall_clock64() - lds.t_start);
  }
}

// After an expand launch in which chains outgrew their working slices: the chains of THAT launch (order[0..n)) whose result
// carries a capacity flag are listed for a relaunch with larger slices — the others keep their packed results — and what
// the listed chains added to the launch's byte counter is taken back (they will add it again).
__global__ void k_collect_overflow(const ChainOut* outs, const uint32_t* order, uint32_t n, uint32_t* redo, uint32_t* n_redo,
                                   Counters* ctr) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  const uint32_t j = order[b];
  const ChainOut co = outs[j];
  if (co.flags & 3u) {  // EG3D_FLAG_CHAIN_OVERFLOW | EG3D_FLAG_OBS_OVERFLOW (include/eg3d.h)
    redo[atomicAdd(n_redo, 1u)] = j;
    if (co.bytes) atomicAdd(&ctr->bytes, 0ull - (unsigned long long)co.bytes);
  }
}

// Cost estimate of a chain for the longest-processing-time-first launch order of K3b:
// initial length x track size of its seed (every track view may attach to every point).
__global__ void k_chain_cost(StageAView a, const TaskDesc* tasks, const ChainSeed* chains, uint32_t n_chains,
                             uint32_t* cost, uint32_t* idx) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_chains) return;
  const ChainSeed cs = chains[j];
  const uint32_t seed = tasks[cs.task].seed;
  const uint32_t k = track_len(a, seed);
  cost[j] = (cs.n1 + 1 + cs.n2) * k;
  idx[j] = j;
}

// ------------------------------------------------------------------ K4 ---------
// One wavefront per chain, in OUTPUT order: the chain's packed record (k3b_expand) becomes its slice
// of the ordered SoA cloud. Lanes take consecutive points; the observation offset of each point is
// the chain's base plus a wave prefix sum of the per-point counts; the observations of the record are
// already one flat range, so lane f copies observation f: coalesced 16-byte reads, coalesced rows in
// every output array.
__global__ void __launch_bounds__(64) k4_emit(const TaskDesc* tasks, const ChainSeed* chains, uint32_t n_chains,
                                              StageBuf stage, const ChainOut* outs, const uint32_t* point_off,
                                              const uint32_t* obs_off_in, uint64_t point_base, uint64_t obs_base,
                                              uint32_t key0_base, float* X, eg3d_off_t* obs_off, int32_t* obs_view,
                                              uint32_t* obs_pl, uint32_t* obs_seg, float* obs_xy, uint32_t* key) {
  const uint32_t j = blockIdx.x;
  const uint32_t lane = threadIdx.x;
  const ChainOut co = outs[j];
  const StagePt* spt = stage.pts + co.spt;
  const Obs* sob = stage.obs + co.sobs;
  const TaskDesc d = tasks[chains[j].task];
  const uint64_t pbase = point_base + point_off[j];
  const uint64_t obase = obs_base + obs_off_in[j];
  uint32_t run = 0;  // observations of the points before this group of 64
  for (uint32_t i0 = 0; i0 < co.n_points; i0 += 64) {
    const uint32_t i = i0 + lane;
    const bool act = i < co.n_points;
    StagePt p;
    p.nobs = 0;
    p.X[0] = p.X[1] = p.X[2] = 0.f;
    if (act) p = spt[i];
    const uint32_t incl = (uint32_t)wave_incl_scan((int)p.nobs);  // inclusive wave scan of the observation counts
    const uint32_t total = lane_bcast(incl, 63);
    if (act) {
      const uint64_t pi = pbase + i;
      X[3 * pi] = p.X[0];
      X[3 * pi + 1] = p.X[1];
      X[3 * pi + 2] = p.X[2];
      obs_off[pi] = (eg3d_off_t)(obase + run + (incl - p.nobs));
      key[4 * pi] = d.seed + key0_base;
      key[4 * pi + 1] = d.entry;
      key[4 * pi + 2] = d.hit;
      key[4 * pi + 3] = i;
    }
    run += total;
  }
  for (uint32_t f = lane; f < co.n_obs; f += 64) {
    const Obs po = sob[f];
    const uint64_t o = obase + f;
    obs_view[o] = po.view;
    obs_pl[o] = po.pl;
    obs_seg[o] = po.seg;
    *(f2*)(obs_xy + 2 * o) = f2{po.x, po.y};
  }
}

// ------------------------------------------------------------------ K5 ---------
// One lane per point, one block per 256 consecutive points. The block's operands are staged in LDS
// with coalesced loads first: the 3x4 part of every camera matrix (V <= 256) and the block's
// observations, which are one contiguous slice of the CSR (<= K5_OBS_CAP of them; k ~ U[3,10] gives
// ~1.5k). A lane then walks ITS list through ds_reads — lane-strided reads of the CSR straight from
// HBM touch ~7x the cache lines they use, and the per-observation camera matrix is a 48-byte gather.
// Blocks whose slice or rig does not fit fall back to the HBM operands (same arithmetic).
// What bounds it is arithmetic, not memory: the filter's convergence test |d mse| < 5e-10 on FP32 values
// of ~0.25 only fires when the mean-square error repeats EXACTLY, so most points run many of the 30
// iterations, each 2 passes x k rows x (8 correctly rounded FP32 divisions + 60 FP64 multiply-adds and
// conversions). Measured alternatives (same bits, slower): persistent lanes with a flattened
// per-row state machine and a global work counter — the end-of-pass step then diverges on almost every
// trip (4.4 ms vs 3.2 ms per 1 M points).
// Round 6 — lane occupancy (tools/c5_iterations.py: on the 1 M-point workload 62 % of the points run all 30 passes, the
// others stop after 4-10, and k spans 3..10; with points taken in input order a wavefront keeps 0.71 of its lanes busy by
// passes and only 0.41 by residual rows, because every pass runs as long as the longest list among its 64 lanes). The
// block's points are therefore SORTED BY k in LDS (counting sort) before lanes take them: a wavefront's lists have (nearly)
// the same length. 2.83 -> 2.17 ms per 1 M points, vector instructions 1.78e9 -> 1.18e9, measured active-lane fraction
// 0.61 (profiles/r06_c5_rocprof_summary.txt). A point's arithmetic does not depend on the lane that runs it, so the output
// is bit for bit what one lane per point in input order produces.
// Measured and dropped: packing the points still running after 6 / 10 passes into the block's first wavefronts (state
// through LDS, order kept; gauss_newton_f32_span makes the iteration resumable): 2.37 / 2.21 ms against 2.17 without —
// a block of 4 wavefronts rarely empties a whole one (0.62 x 256 points = 2.5 wavefronts), and the freed lanes of a
// partly empty wavefront cost nothing extra.
#define K5_BLOCK 256
#define K5_OBS_CAP 2816
#define K5_VIEW_CAP 256
#define K5_KBUCKETS 65 /* list lengths 0..63 and "64 or more" */
__global__ void __launch_bounds__(K5_BLOCK) k5_gn_filter(const float* cam_P, int n_views, const float* X,
                                                        const uint32_t* obs_off, const int32_t* obs_view,
                                                        const float* obs_xy, uint64_t n, float gn_max_mse,
                                                        int legacy_abs, float* X_out, uint8_t* inlier) {
  typedef const __attribute__((address_space(3))) float* lds_fp;
  typedef const __attribute__((address_space(3))) int32_t* lds_ip;
  __shared__ float sP[K5_VIEW_CAP * 12];
  __shared__ int32_t sV[K5_OBS_CAP];
  __shared__ float sXY[2 * K5_OBS_CAP];
  __shared__ uint32_t s_cnt[K5_KBUCKETS], s_base[K5_KBUCKETS];
  __shared__ uint16_t s_perm[K5_BLOCK];
  const uint64_t p0 = (uint64_t)blockIdx.x * K5_BLOCK;
  const uint64_t p1 = p0 + K5_BLOCK < n ? p0 + K5_BLOCK : n;
  const uint32_t np = (uint32_t)(p1 - p0);
  const uint32_t o0 = obs_off[p0], o1 = obs_off[p1];
  const uint32_t m = o1 - o0;
  const uint32_t t = threadIdx.x;
  const bool staged = n_views <= K5_VIEW_CAP && m <= K5_OBS_CAP;  // block-uniform
  if (staged) {
    for (uint32_t q = t; q < (uint32_t)n_views * 12u; q += K5_BLOCK) sP[q] = cam_P[(q / 12u) * 16u + q % 12u];
    for (uint32_t q = t; q < m; q += K5_BLOCK) sV[q] = obs_view[o0 + q];
    for (uint32_t q = t; q < 2u * m; q += K5_BLOCK) sXY[q] = obs_xy[2 * (size_t)o0 + q];
  }
  if (t < K5_KBUCKETS) s_cnt[t] = 0;
  __syncthreads();
  // ---- (1) counting sort of the block's points by list length (the order inside a bucket is whatever the LDS atomics
  // give: it decides which lane runs a point, not what the point computes)
  uint32_t kb = 0, rank = 0;
  if (t < np) {
    const uint32_t k = obs_off[p0 + t + 1] - obs_off[p0 + t];
    kb = k < K5_KBUCKETS - 1 ? k : K5_KBUCKETS - 1;
    rank = atomicAdd(&s_cnt[kb], 1u);
  }
  __syncthreads();
  if (t < K5_KBUCKETS) {
    uint32_t base = 0;
    for (uint32_t q = 0; q < t; q++) base += s_cnt[q];
    s_base[t] = base;
  }
  __syncthreads();
  if (t < np) s_perm[s_base[kb] + rank] = (uint16_t)t;
  __syncthreads();
  // ---- lane t takes the t-th point of the sorted order
  auto run_span = [&](uint32_t j, GnF32State& st, int it0, int it1) -> int {
    const uint64_t i = p0 + j;
    const uint32_t a = obs_off[i], b = obs_off[i + 1];
    if (staged)
      return gauss_newton_f32_span((lds_fp)&sP[0], 12, (lds_ip)&sV[0] + (a - o0), (lds_fp)&sXY[0] + 2 * (a - o0), (int)(b - a),
                                   st, legacy_abs != 0, it0, it1);
    return gauss_newton_f32_span(cam_P, 16, obs_view + a, obs_xy + 2 * (size_t)a, (int)(b - a), st, legacy_abs != 0, it0, it1);
  };
  auto finish = [&](uint32_t j, const GnF32State& st, int r) {  // gauss_newton.cpp:130-133: accepted on the last mse, converged or not
    const uint64_t i = p0 + j;
    const bool ok = r != GN_F32_FAILED && st.last_mse < gn_max_mse;
    const float x0 = X[3 * i], x1 = X[3 * i + 1], x2 = X[3 * i + 2];
    inlier[i] = ok ? 1 : 0;
    X_out[3 * i] = ok ? st.X[0] : x0;
    X_out[3 * i + 1] = ok ? st.X[1] : x1;
    X_out[3 * i + 2] = ok ? st.X[2] : x2;
  };
  if (t < np) {
    const uint32_t j = s_perm[t];
    const uint64_t i = p0 + j;
    GnF32State st;
    st.X[0] = X[3 * i];
    st.X[1] = X[3 * i + 1];
    st.X[2] = X[3 * i + 2];
    st.last_mse = 0;
    finish(j, st, run_span(j, st, 0, 30));
  }
}

// Exclusive scans of 32-bit counts wrap silently when the total passes 2^32. The counts are
// non-negative, so a wrapped scan is exactly one whose output decreases somewhere: this check runs
// after every scan and ORs into a device flag word that k_publish hands to the host (and clears).
__global__ void k_scan_check(const uint32_t* out, uint64_t n_plus_one, uint32_t* wrapped) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i + 1 < n_plus_one && out[i + 1] < out[i]) atomicOr(wrapped, 1u);
}
void launch_scan_check(hipStream_t st, const uint32_t* out, uint64_t n_plus_one, uint32_t* wrapped) {
  hipLaunchKernelGGL(k_scan_check, dim3((unsigned)((n_plus_one + 255) / 256)), dim3(256), 0, st, out, n_plus_one, wrapped);
}

// Small device -> host read-backs (scan totals, counters) without a driver round trip: one wavefront
// copies the listed device words into a mailbox in pinned, GPU-mapped host memory and then stores
// the sequence number the host thread is polling for (system-scope release after system fences).
__global__ void __launch_bounds__(64) k_publish(PubArgs a, uint32_t* mbox, uint32_t seq) {
  uint32_t at = 2;  // [0] sequence number, [1] unused, payload from [2]
  for (int i = 0; i < a.n; i++) {
    for (uint32_t w = threadIdx.x; w < a.words[i]; w += 64) mbox[at + w] = a.src[i][w];
    at += a.words[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < a.n_clear; i++) *a.clear[i] = 0;
    __threadfence_system();
    __hip_atomic_store(mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
void launch_publish(hipStream_t st, const PubArgs& a, uint32_t* mbox_dev, uint32_t seq) {
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, a, mbox_dev, seq);
}

// ------------------------------------------------------------ launch wrappers --
static inline dim3 blocks_for(uint64_t n, uint32_t per_block) { return dim3((unsigned)((n + per_block - 1) / per_block)); }

void launch_seed_prep(hipStream_t st, SeedsDev sd, uint32_t seed_begin, uint32_t n_seeds, uint32_t sv_base,
                      uint32_t* sv_seed, int32_t* map_view, uint32_t* map_entry, uint32_t* map_n) {
  if (!n_seeds) return;
  hipLaunchKernelGGL(k_seed_prep, blocks_for(n_seeds, 256), dim3(256), 0, st, sd, seed_begin, n_seeds, sv_base, sv_seed,
                     map_view, map_entry, map_n);
}
void launch_k1_count_raw(hipStream_t st, DevScene s, SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
                         uint32_t* raw_cnt) {
  if (!n_sv) return;
  hipLaunchKernelGGL(k1_count_raw, blocks_for(n_sv, 256), dim3(256), 0, st, s, sd, sv_base, n_sv, sv_seed, raw_cnt);
}
void launch_k1(hipStream_t st, DevScene s, SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
               const uint32_t* raw_off, uint32_t* cand_pl, Obs* start_hits, uint32_t* cand_cnt, uint32_t* start_cnt,
               uint32_t* sv_vtx) {
  if (!n_sv) return;
  hipLaunchKernelGGL(k1_seed_candidates, blocks_for((uint64_t)n_sv * 64, 256), dim3(256), 0, st, s, sd, sv_base, n_sv,
                     sv_seed, raw_off, cand_pl, start_hits, cand_cnt, start_cnt, sv_vtx);
}
void launch_task_fill(hipStream_t st, SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
                      const uint32_t* start_cnt, const uint32_t* task_off, uint32_t* task_seed, uint32_t* task_entry,
                      uint32_t* task_hit, uint32_t* task_k, const uint32_t* sv_vtx, Counters* ctr) {
  if (!n_sv) return;
  hipLaunchKernelGGL(k_task_fill, blocks_for(n_sv, 256), dim3(256), 0, st, sd, sv_base, n_sv, sv_seed, start_cnt,
                     task_off, task_seed, task_entry, task_hit, task_k, sv_vtx, ctr);
}
void launch_k2(hipStream_t st, bool fill, DevScene s, SeedsDev sd, uint32_t seed_begin, uint32_t n_seeds, uint32_t sv_base,
               uint32_t n_tasks, const uint32_t* task_off, const uint32_t* task_seed, const uint32_t* task_entry,
               const uint32_t* task_hit, const uint32_t* task_list_off, const uint32_t* raw_off, const uint32_t* cand_pl,
               const uint32_t* cand_cnt, const Obs* start_hits, uint32_t* list_cnt, const uint32_t* list_ptr, Obs* hits) {
  if (!n_tasks || !n_seeds) return;
  if (fill)
    hipLaunchKernelGGL(k2_epipolar_hits<true>, blocks_for((uint64_t)n_tasks * 64, 256), dim3(256), 0, st, s, sd, sv_base,
                       n_tasks, task_seed, task_entry, task_hit, task_list_off, raw_off, cand_pl, cand_cnt, start_hits,
                       list_cnt, list_ptr, hits);
  else
    hipLaunchKernelGGL(k2_epipolar_hits<false>, blocks_for((uint64_t)n_tasks * 64, 256), dim3(256), 0, st, s, sd,
                       sv_base, n_tasks, task_seed, task_entry, task_hit, task_list_off, raw_off, cand_pl, cand_cnt,
                       start_hits, list_cnt, list_ptr, hits);
}
void launch_n1_samples(hipStream_t st, bool fill, DevScene s, SetsDev sets, uint32_t n_rows, uint32_t item_begin,
                       uint32_t n_items, uint32_t* sample_cnt, const uint32_t* sample_off, Obs* samples,
                       uint32_t* task_seed, uint32_t* task_entry, uint32_t* task_hit, uint32_t* task_list_off,
                       uint32_t* task_row0, Counters* ctr) {
  if (!n_items) return;
  if (fill)
    hipLaunchKernelGGL(k_n1_samples<true>, blocks_for(n_items, 64), dim3(64), 0, st, s, sets, n_rows, item_begin,
                       n_items, sample_cnt, sample_off, samples, task_seed, task_entry, task_hit, task_list_off,
                       task_row0, ctr);
  else
    hipLaunchKernelGGL(k_n1_samples<false>, blocks_for(n_items, 64), dim3(64), 0, st, s, sets, n_rows, item_begin,
                       n_items, sample_cnt, sample_off, samples, task_seed, task_entry, task_hit, task_list_off,
                       task_row0, ctr);
}
void launch_n1_hits(hipStream_t st, bool fill, DevScene s, SetsDev sets, uint32_t n_tasks, const Obs* samples,
                    const uint32_t* task_row0, uint32_t* list_cnt, const uint32_t* list_ptr, Obs* hits, Counters* ctr) {
  if (!n_tasks) return;
  if (fill)
    hipLaunchKernelGGL(k_n1_hits<true>, blocks_for(n_tasks, 256), dim3(256), 0, st, s, sets, n_tasks, samples, task_row0,
                       list_cnt, list_ptr, hits, ctr);
  else
    hipLaunchKernelGGL(k_n1_hits<false>, blocks_for(n_tasks, 256), dim3(256), 0, st, s, sets, n_tasks, samples, task_row0,
                       list_cnt, list_ptr, hits, ctr);
}
void launch_task_setup(hipStream_t st, StageAView a, const int32_t* map_view, const uint32_t* map_entry,
                       const uint32_t* map_n, TaskDesc* tasks, uint32_t* n_hyp) {
  if (!a.n_tasks) return;
  hipLaunchKernelGGL(k_task_setup, blocks_for(a.n_tasks, 256), dim3(256), 0, st, a, map_view, map_entry, map_n, tasks,
                     n_hyp);
}
// K3a as a request/serve engine (eg3d_k3a_engine.h): one wavefront per block
void launch_k3a_engine(hipStream_t st, uint32_t orient_waves, uint32_t follow_waves, uint32_t lanes_per_wave, DevScene s,
                       StageAView a, const TaskDesc* tasks, const uint32_t* hyp_off, uint32_t n_hyp, HypResult* res,
                       HPoint* follow_scratch, uint32_t hyp_cap, HPoint* arena, uint32_t arena_cap, Counters* ctr,
                       uint32_t* queue3, uint32_t* items) {
  if (!n_hyp) return;
  hipLaunchKernelGGL(k3a_orient, dim3(orient_waves), dim3(64), 0, st, s, a, tasks, hyp_off, n_hyp, res, hyp_cap, arena,
                     arena_cap, ctr, queue3, lanes_per_wave, items, queue3 + 2);
  hipLaunchKernelGGL(k3a_follow_spec, dim3(follow_waves), dim3(64), 0, st, s, res, follow_scratch, hyp_cap, arena, arena_cap, ctr, queue3 + 1, lanes_per_wave, items, queue3 + 2);
  hipLaunchKernelGGL(k3a_finalize, blocks_for(n_hyp, 256), dim3(256), 0, st, n_hyp, res);
}
void launch_k3s(hipStream_t st, uint32_t n_tasks, const uint32_t* hyp_off, const HypResult* res, ChainSeed* per_task,
                uint32_t* valid) {
  if (!n_tasks) return;
  hipLaunchKernelGGL(k3s_select, blocks_for(n_tasks, 256), dim3(256), 0, st, n_tasks, hyp_off, res, per_task, valid);
}
void launch_compact_chains(hipStream_t st, uint32_t n_tasks, const ChainSeed* per_task, const uint32_t* valid,
                           const uint32_t* chain_off, ChainSeed* chains) {
  if (!n_tasks) return;
  hipLaunchKernelGGL(k_compact_chains, blocks_for(n_tasks, 256), dim3(256), 0, st, n_tasks, per_task, valid, chain_off,
                     chains);
}
#ifdef EG3D_GN_COUNTERS
int gn_dbg_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gn_dbg), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[128] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gn_dbg), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
// Three instantiations (round 4), see TeamWaveT: small scenes (50.6 k -> 45 k instructions, 116 instead of 262 spilled
// vector registers), general, many views.
static_assert(EG3D_STAGE_VTX_HOST == EG3D_STAGE_VTX, "the host's small-scene rule must match the side walks' staging capacity");
static_assert(EG3D_GN_PACK_MAX >= EG3D_SMALL_SCENE_VIEWS_HOST, "a small scene's solves must fit a packed round");
static_assert(2 * EG3D_SMALL_SCENE_VIEWS_HOST + 8 <= EG3D_COOP_ROWS && 2 * (EG3D_SMALL_SCENE_VIEWS_HOST + 1) + 8 > EG3D_COOP_ROWS,
              "EG3D_SMALL_SCENE_VIEWS_HOST must be the last view count whose N-view step lists fit LDS");
static constexpr auto k3b_expand_small = k3b_expand_t<EG3D_K3B_WAVES, 0, 0>;
static constexpr auto k3b_expand = k3b_expand_t<EG3D_K3B_WAVES, 0, 1>;
#ifndef EG3D_MANY_KEEP
#define EG3D_MANY_KEEP 1 /* chunks of a long solve whose rows stay in registers between the passes of an iteration, many-views build (gn_round<KEEP>): 0 / 1 / 2 -> C4 step 1810 / 1774 / 1823 ms at 27 / 60 / 263 spilled VGPRs (round 5; round 4 had only measured 4 chunks at 2 waves per SIMD: slower) */
#endif
#ifndef EG3D_MANY_WAVES
#define EG3D_MANY_WAVES EG3D_K3B_WAVES /* waves per SIMD of the many-views build (fewer = more registers for kept chunks) */
#endif
static constexpr auto k3b_expand_many = k3b_expand_t<EG3D_MANY_WAVES, EG3D_MANY_KEEP, 2>;
int k3b_blocks_per_cu() {  // the largest residency of the builds sizes the slot pools
  int best = 0;
  for (int k = 0; k < 3; k++) {
    int n = 0;
    const void* f = k == 0 ? (const void*)k3b_expand_small : k == 1 ? (const void*)k3b_expand : (const void*)k3b_expand_many;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 64, 0) != hipSuccess || n < 1) return 0;
    best = n > best ? n : best;
  }
  return best;
}
void launch_pool_init(hipStream_t st, SlotPools pools) {
  hipLaunchKernelGGL(k_pool_init, dim3(8), dim3(256), 0, st, pools);
}
void launch_k3b(hipStream_t st, DevScene s, StageAView a, const TaskDesc* tasks, const ChainSeed* chains,
                uint32_t n_chains, const uint32_t* hyp_off, const HypResult* res, const HPoint* arena,
                const int32_t* map_view, const uint32_t* map_entry, const uint32_t* map_n, ChainLayout L,
                unsigned char* slices, SlotPools pools, StageBuf stage, ChainOut* outs, uint32_t* out_points,
                uint32_t* out_obs, Counters* ctr, const uint32_t* order, int scene_class) {
  if (!n_chains) return;
  if (scene_class == 0)
    hipLaunchKernelGGL(k3b_expand_small, dim3(n_chains), dim3(64), 0, st, s, a, tasks, chains, n_chains, hyp_off, res,
                       arena, map_view, map_entry, map_n, L, slices, pools, stage, outs, out_points, out_obs, ctr, order);
  else if (scene_class == 2)
    hipLaunchKernelGGL(k3b_expand_many, dim3(n_chains), dim3(64), 0, st, s, a, tasks, chains, n_chains, hyp_off, res,
                       arena, map_view, map_entry, map_n, L, slices, pools, stage, outs, out_points, out_obs, ctr, order);
  else
    hipLaunchKernelGGL(k3b_expand, dim3(n_chains), dim3(64), 0, st, s, a, tasks, chains, n_chains, hyp_off, res,
                       arena, map_view, map_entry, map_n, L, slices, pools, stage, outs, out_points, out_obs, ctr, order);
}
}  // namespace eg3d
// The lane-per-chain engine is a second, slower form of the expand stage: compiled only into builds made with
// -DEG3D_WITH_K3C_ENGINE (edgegraph3d_amd/build.py build_hip_engine -> variants/libeg3d_engine.so); the product libraries
// do not carry it. Its state machine (eg3d_chain_sm.h) stays checked on the host by tests/hostsim.
#ifdef EG3D_WITH_K3C_ENGINE
#include "eg3d_k3c_engine.h"
#endif
namespace eg3d {
#ifdef EG3D_WITH_K3C_ENGINE
// The expand stage as a lane-per-chain engine (eg3d_k3c_engine.h): n_waves single-wavefront blocks whose first
// lanes_per_wave lanes each own a working slice (slices: [n_waves * lanes_per_wave] x L.total bytes) and take chains from
// the launch's queue (*queue zeroed by the caller) until it is empty. long_gn = 0: scenes whose solves all fit a packed
// round of the solver (<= 32 rows); a longer request raises CTR_LONG_REFUSED and the host repeats the launch with 1.
static constexpr auto k3c_engine_short = k3c_engine_t<false>;
static constexpr auto k3c_engine_long = k3c_engine_t<true>;
int k3c_blocks_per_cu() {
  int a = 0, b = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, (const void*)k3c_engine_short, 64, 0) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, (const void*)k3c_engine_long, 64, 0) != hipSuccess) return 0;
  return a < b ? a : b;
}
void launch_k3c(hipStream_t st, uint32_t n_waves, uint32_t lanes_per_wave, DevScene s, StageAView a, const TaskDesc* tasks,
                const ChainSeed* chains, uint32_t n_chains, const uint32_t* hyp_off, const HypResult* res, const HPoint* arena,
                const int32_t* map_view, const uint32_t* map_entry, const uint32_t* map_n, ChainLayout L, unsigned char* slices,
                StageBuf stage, ChainOut* outs, uint32_t* out_points, uint32_t* out_obs, Counters* ctr, const uint32_t* order,
                uint32_t* queue, int long_gn) {
  if (!n_chains) return;
  if (long_gn)
    hipLaunchKernelGGL(k3c_engine_long, dim3(n_waves), dim3(64), 0, st, s, a, tasks, chains, n_chains, hyp_off, res, arena,
                       map_view, map_entry, map_n, L, slices, stage, outs, out_points, out_obs, ctr, order, queue, lanes_per_wave);
  else
    hipLaunchKernelGGL(k3c_engine_short, dim3(n_waves), dim3(64), 0, st, s, a, tasks, chains, n_chains, hyp_off, res, arena,
                       map_view, map_entry, map_n, L, slices, stage, outs, out_points, out_obs, ctr, order, queue, lanes_per_wave);
}
#endif  // EG3D_WITH_K3C_ENGINE
#if defined(EG3D_WITH_K3C_ENGINE) && (defined(EG3D_SECTION_TIMING) || defined(EG3D_K3C_TIMING))
int k3c_dbg_read(unsigned long long* out, int reset) {  // out[128]: g_k3c_dbg[32] then g_k3c_prof[96]
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k3c_dbg), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out + 32, HIP_SYMBOL(g_k3c_prof), sizeof(unsigned long long) * 96) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[96] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_k3c_dbg), z, sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_k3c_prof), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
void launch_k0_pairs(hipStream_t st, bool fill, DevScene s, uint32_t n_pl, float cell_dim, int map_w, int map_h, uint32_t* cnt,
                     const uint32_t* off, unsigned long long* keys, uint32_t* dropped) {
  if (!n_pl) return;
  if (fill)
    hipLaunchKernelGGL(k0_grid_pairs<true>, blocks_for(n_pl, 64), dim3(64), 0, st, s, n_pl, cell_dim, map_w, map_h, cnt, off, keys, dropped);
  else
    hipLaunchKernelGGL(k0_grid_pairs<false>, blocks_for(n_pl, 64), dim3(64), 0, st, s, n_pl, cell_dim, map_w, map_h, cnt, off, keys, dropped);
}
void launch_k0_csr(hipStream_t st, const unsigned long long* keys, uint32_t n, uint32_t total_cells, uint32_t* off, uint32_t* ids) {
  if (!n) return;
  hipLaunchKernelGGL(k0_grid_csr, blocks_for(n, 256), dim3(256), 0, st, keys, n, total_cells, off, ids);
}
void launch_collect_overflow(hipStream_t st, const ChainOut* outs, const uint32_t* order, uint32_t n, uint32_t* redo,
                             uint32_t* n_redo, Counters* ctr) {
  if (!n) return;
  hipLaunchKernelGGL(k_collect_overflow, blocks_for(n, 256), dim3(256), 0, st, outs, order, n, redo, n_redo, ctr);
}
void launch_chain_cost(hipStream_t st, StageAView a, const TaskDesc* tasks, const ChainSeed* chains, uint32_t n_chains,
                       uint32_t* cost, uint32_t* idx) {
  if (!n_chains) return;
  hipLaunchKernelGGL(k_chain_cost, blocks_for(n_chains, 256), dim3(256), 0, st, a, tasks, chains, n_chains, cost, idx);
}
void launch_k4(hipStream_t st, const TaskDesc* tasks, const ChainSeed* chains, uint32_t n_chains, StageBuf stage,
               const ChainOut* outs, const uint32_t* point_off, const uint32_t* obs_off_in, uint64_t point_base,
               uint64_t obs_base, uint32_t key0_base, float* X, eg3d_off_t* obs_off, int32_t* obs_view, uint32_t* obs_pl,
               uint32_t* obs_seg, float* obs_xy, uint32_t* key) {
  if (!n_chains) return;
  hipLaunchKernelGGL(k4_emit, dim3(n_chains), dim3(64), 0, st, tasks, chains, n_chains, stage, outs, point_off,
                     obs_off_in, point_base, obs_base, key0_base, X, obs_off, obs_view, obs_pl, obs_seg, obs_xy, key);
}
void launch_k5(hipStream_t st, const float* cam_P, int n_views, const float* X, const uint32_t* obs_off,
               const int32_t* obs_view, const float* obs_xy, uint64_t n, float gn_max_mse, int legacy_abs, float* X_out,
               uint8_t* inlier) {
  if (!n) return;
  hipLaunchKernelGGL(k5_gn_filter, blocks_for(n, K5_BLOCK), dim3(K5_BLOCK), 0, st, cam_P, n_views, X, obs_off, obs_view,
                     obs_xy, n, gn_max_mse, legacy_abs, X_out, inlier);
}

}  // namespace eg3d
