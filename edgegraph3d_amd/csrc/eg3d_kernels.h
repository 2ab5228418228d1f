// Host-visible declarations of the kernel launch wrappers (eg3d_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eg3d_dev_pipeline.h"

namespace eg3d {

struct SeedsDev {
  const uint32_t* trk_off;
  const int32_t* trk_view;
  const float* trk_xy;
};

enum : uint32_t { CTR_ARENA_OVERFLOW = 0x100u, CTR_SLOT_STARVED = 0x200u /* k3b_expand found no free working slice (internal) */,
                  CTR_LONG_REFUSED = 0x400u /* the few-views build of k3b_expand met a solve of more than 32 rows (internal) */ };
typedef uint64_t eg3d_off_t;  // element type of obs_off in the output cloud (include/eg3d.h)
// arguments of k_publish: up to 6 runs of device words copied to the host mailbox, words cleared afterwards
struct PubArgs {
  const uint32_t* src[6];
  uint32_t words[6];
  uint32_t* clear[3];
  int n, n_clear;
};
enum { EG3D_MBOX_WORDS = 64 };
void launch_publish(hipStream_t st, const PubArgs& a, uint32_t* mbox_dev, uint32_t seq);
struct Counters {
  uint32_t arena_used;
  uint32_t flags;  // EG3D_FLAG_* bits | CTR_ARENA_OVERFLOW
  unsigned long long bytes;  // algorithmic bytes of polyline vertices touched
  uint32_t max_chain_ticks;  // k3b_expand: the longest time one chain held its wavefront (ticks of the constant-rate clock, wall_clock64)
  uint32_t pad_;
};

void launch_seed_prep(hipStream_t st, SeedsDev sd, uint32_t seed_begin, uint32_t n_seeds, uint32_t sv_base,
                      uint32_t* sv_seed, int32_t* map_view, uint32_t* map_entry, uint32_t* map_n);
void launch_k1_count_raw(hipStream_t st, DevScene s, SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
                         uint32_t* raw_cnt);
void launch_k1(hipStream_t st, DevScene s, SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
               const uint32_t* raw_off, uint32_t* cand_pl, Obs* start_hits, uint32_t* cand_cnt, uint32_t* start_cnt,
               uint32_t* sv_vtx);
void launch_task_fill(hipStream_t st, SeedsDev sd, uint32_t sv_base, uint32_t n_sv, const uint32_t* sv_seed,
                      const uint32_t* start_cnt, const uint32_t* task_off, uint32_t* task_seed, uint32_t* task_entry,
                      uint32_t* task_hit, uint32_t* task_k, const uint32_t* sv_vtx, Counters* ctr);
#ifndef EG3D_K2_LDS
#define EG3D_K2_LDS 0
#endif
// EG3D_K2_LDS=0 (default): one wavefront per task. =1: one workgroup per seed of [seed_begin, seed_begin + n_seeds)
// with the candidate polylines staged in LDS; task_off = first task of each (seed, track entry).
void launch_k2(hipStream_t st, bool fill, DevScene s, SeedsDev sd, uint32_t seed_begin, uint32_t n_seeds, uint32_t sv_base,
               uint32_t n_tasks, const uint32_t* task_off, const uint32_t* task_seed, const uint32_t* task_entry,
               const uint32_t* task_hit, const uint32_t* task_list_off, const uint32_t* raw_off, const uint32_t* cand_pl,
               const uint32_t* cand_cnt, const Obs* start_hits, uint32_t* list_cnt, const uint32_t* list_ptr, Obs* hits);
// ---- pipelines 1-2 extractor: sets of potentially compatible polylines -> tasks ----
// CSR over rows (set * V + view) of view-local polyline ids, ascending per row (device copies).
struct SetsDev {
  uint32_t n_views;
  const uint32_t* row_off;  // [n_sets * V + 1]
  const uint32_t* pl_ids;
};
// items = entries [item_begin, item_begin + n_items) of pl_ids; fill=false counts the 20 px samples
// of each item's polyline, fill=true writes them (one task per sample) at sample_off[item]
void launch_n1_samples(hipStream_t st, bool fill, DevScene s, SetsDev sets, uint32_t n_rows, uint32_t item_begin,
                       uint32_t n_items, uint32_t* sample_cnt, const uint32_t* sample_off, Obs* samples,
                       uint32_t* task_seed, uint32_t* task_entry, uint32_t* task_hit, uint32_t* task_list_off,
                       uint32_t* task_row0, Counters* ctr);
// one wavefront per (task, view): hits of the sample's epipolar line on the set's polylines of that view
void launch_n1_hits(hipStream_t st, bool fill, DevScene s, SetsDev sets, uint32_t n_tasks, const Obs* samples,
                    const uint32_t* task_row0, uint32_t* list_cnt, const uint32_t* list_ptr, Obs* hits, Counters* ctr);
void launch_task_setup(hipStream_t st, StageAView a, const int32_t* map_view, const uint32_t* map_entry,
                       const uint32_t* map_n, TaskDesc* tasks, uint32_t* n_hyp);
// K3a as a request/serve engine (eg3d_k3a_engine.h): orient_waves / follow_waves single-wavefront blocks, of which
// the first lanes_per_wave lanes take work; follow_scratch: min(hyp_cap, EG3D_K3A_STAGE_POINTS) HPoints per lane of
// follow_waves x 64; queue3: three
// zeroed counters (hypotheses taken, lists taken, lists to follow); items: 2 x n_hyp words (the lists to follow)
#define EG3D_K3A_STAGE_POINTS 16u /* = EG3D_K3A_STAGE of eg3d_k3a_engine.h */
void launch_k3a_engine(hipStream_t st, uint32_t orient_waves, uint32_t follow_waves, uint32_t lanes_per_wave, DevScene s,
                       StageAView a, const TaskDesc* tasks, const uint32_t* hyp_off, uint32_t n_hyp, HypResult* res,
                       HPoint* follow_scratch, uint32_t hyp_cap, HPoint* arena, uint32_t arena_cap, Counters* ctr,
                       uint32_t* queue3, uint32_t* items);
void launch_k3s(hipStream_t st, uint32_t n_tasks, const uint32_t* hyp_off, const HypResult* res, ChainSeed* per_task,
                uint32_t* valid);
void launch_compact_chains(hipStream_t st, uint32_t n_tasks, const ChainSeed* per_task, const uint32_t* valid,
                           const uint32_t* chain_off, ChainSeed* chains);
// Pools of working-slice slots of k3b_expand, one per XCD (see the kernel): 8 x `stride` words,
// [0] pop tickets, [16] push tickets, [32 ..] a ring of ring_n (power of two >= slots_per_xcd) cells.
struct SlotPools {
  uint32_t* base;
  uint32_t stride, ring_n, slots_per_xcd;
};
#define EG3D_SMALL_SCENE_VIEWS_HOST 28 /* the last view count whose N-view step lists fit LDS (2 V + 8 <= 64; static_assert in eg3d_kernels.hip) */
#define EG3D_STAGE_VTX_HOST 512 /* = EG3D_STAGE_VTX: vertices of a polyline the side walks stage in LDS */
int k3b_blocks_per_cu();  // resident k3b_expand workgroups per CU (occupancy query; 0 on failure)
void launch_pool_init(hipStream_t st, SlotPools pools);
// lane-per-chain engine of the expand stage (eg3d_k3c_engine.h; builds with -DEG3D_WITH_K3C_ENGINE only); see eg3d_kernels.hip
#ifdef EG3D_WITH_K3C_ENGINE
int k3c_blocks_per_cu();
void launch_k3c(hipStream_t st, uint32_t n_waves, uint32_t lanes_per_wave, DevScene s, StageAView a, const TaskDesc* tasks,
                const ChainSeed* chains, uint32_t n_chains, const uint32_t* hyp_off, const HypResult* res, const HPoint* arena,
                const int32_t* map_view, const uint32_t* map_entry, const uint32_t* map_n, ChainLayout L, unsigned char* slices,
                StageBuf stage, ChainOut* outs, uint32_t* out_points, uint32_t* out_obs, Counters* ctr, const uint32_t* order,
                uint32_t* queue, int long_gn);
#endif
void launch_k3b(hipStream_t st, DevScene s, StageAView a, const TaskDesc* tasks, const ChainSeed* chains,
                uint32_t n_chains, const uint32_t* hyp_off, const HypResult* res, const HPoint* arena,
                const int32_t* map_view, const uint32_t* map_entry, const uint32_t* map_n, ChainLayout L,
                unsigned char* slices, SlotPools pools, StageBuf stage, ChainOut* outs, uint32_t* out_points,
                uint32_t* out_obs, Counters* ctr, const uint32_t* order, int scene_class /* polylines of <= 512 vertices: 0 small (<= 28 views) or 2 many views (>= 29); 1 general: the build of the kernel without the paths such a scene cannot reach */);
#define EG3D_K0_PL_BITS_HOST 19 /* = EG3D_K0_PL_BITS (static_assert in eg3d_kernels.hip) */
// K0: the uniform grids on the device (eg3d_kernels.hip). fill = false counts the (cell, polyline) pairs of every polyline
// into cnt[g]; fill = true writes them as 64-bit keys (view * cells + cell) << 19 | polyline behind off[g].
void launch_k0_pairs(hipStream_t st, bool fill, DevScene s, uint32_t n_pl, float cell_dim, int map_w, int map_h, uint32_t* cnt,
                     const uint32_t* off, unsigned long long* keys, uint32_t* dropped);
void launch_k0_csr(hipStream_t st, const unsigned long long* keys, uint32_t n, uint32_t total_cells, uint32_t* off, uint32_t* ids);
void launch_collect_overflow(hipStream_t st, const ChainOut* outs, const uint32_t* order, uint32_t n, uint32_t* redo,
                             uint32_t* n_redo, Counters* ctr);
void launch_chain_cost(hipStream_t st, StageAView a, const TaskDesc* tasks, const ChainSeed* chains, uint32_t n_chains,
                       uint32_t* cost, uint32_t* idx);
void launch_k4(hipStream_t st, const TaskDesc* tasks, const ChainSeed* chains, uint32_t n_chains, StageBuf stage,
               const ChainOut* outs, const uint32_t* point_off, const uint32_t* obs_off_in, uint64_t point_base,
               uint64_t obs_base, uint32_t key0_base, float* X, eg3d_off_t* obs_off, int32_t* obs_view, uint32_t* obs_pl,
               uint32_t* obs_seg, float* obs_xy, uint32_t* key);
// {out[n], 1 if the scan wrapped} -> total_and_flag[0..1] (flag word must be zero before the launch)
void launch_scan_check(hipStream_t st, const uint32_t* out, uint64_t n_plus_one, uint32_t* wrapped);
void launch_k5(hipStream_t st, const float* cam_P, int n_views, const float* X, const uint32_t* obs_off,
               const int32_t* obs_view, const float* obs_xy, uint64_t n, float gn_max_mse, int legacy_abs, float* X_out,
               uint8_t* inlier);

}  // namespace eg3d
