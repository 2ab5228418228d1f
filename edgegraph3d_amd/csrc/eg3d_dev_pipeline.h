// eg3d_dev_pipeline.h — per-work-item bodies of the phase pipeline
//   K1 seed_candidates -> K2 epipolar_hits -> T task_setup -> K3a hypotheses ->
//   K3s select -> K3b expand -> K4 emit
// Each function is the body one GPU lane runs for one item; the __global__ wrappers in
// eg3d_kernels.hip only map thread ids to items. (The test-only host simulation under
// tests/hostsim drives the same bodies serially to debug logic without a GPU.)
//
// Reference behaviour covered here: the 3-view selection and uniqueness rule of
// compute_3D_point_multiple_views_plg_following_expandallviews_vector and
// compute_unique_potential_3d_points_3views_... (triangulation.cpp:1027-1088, 550-601,
// including the stale direction-2 buffer, Q12), the view-indexed scatter of
// consensus_strategy_single_point_single_intersection (plgpcm_3views_plg_following.cpp:40-50)
// and the emission order of plg_matching_from_refpoint (plg_matching_from_refpoints.cpp:64-81).
#pragma once
#include "eg3d_dev_expand.h"

namespace eg3d {

// ---- stage A products as consumed by stage B (all in HBM) ----
struct StageAView {
  const uint32_t* trk_off;   // [N+1] (whole seed set)
  const int32_t* trk_view;
  const float* trk_xy;
  uint32_t seed_begin;       // first seed of this batch
  uint32_t sv_base;          // trk_off[seed_begin]
  // per task
  uint32_t n_tasks;
  const uint32_t* task_seed;      // [n_tasks]
  const uint32_t* task_entry;     // [n_tasks] track entry index of the start view
  const uint32_t* task_hit;       // [n_tasks] start hit index within (seed, entry)
  const uint32_t* task_list_off;  // [n_tasks+1] one list per track entry of the seed
  const uint32_t* list_ptr;       // [n_lists] offset into hits
  const uint32_t* list_cnt;       // [n_lists]
  const Obs* hits;                // epipolar hits (view filled in)
  // Dense mode (pipelines 1-2 extractor): every "seed" is a sampled polyline point whose track is
  // the identity over all dense_k = V views (entry j = view j, one list per view); no per-seed
  // track or view-map arrays exist — map_view/map_entry hold ONE identity row of dense_k entries.
  uint32_t dense_k = 0;
};
EG3D_HD uint32_t track_base(const StageAView& a, uint32_t seed) { return a.dense_k ? 0u : a.trk_off[seed] - a.sv_base; }
EG3D_HD uint32_t track_len(const StageAView& a, uint32_t seed) {
  return a.dense_k ? a.dense_k : a.trk_off[seed + 1] - a.trk_off[seed];
}
EG3D_HD int32_t track_view(const StageAView& a, uint32_t seed, uint32_t entry) {
  return a.dense_k ? (int32_t)entry : a.trk_view[a.trk_off[seed] + entry];
}
EG3D_HD uint32_t track_n_views(const StageAView& a, const uint32_t* map_n, uint32_t seed) {
  return a.dense_k ? a.dense_k : map_n[seed - a.seed_begin];
}

// view-indexed scatter of a seed's track: distinct views ascending, last entry wins (Q2-like,
// plgpcm_3views_plg_following.cpp:42-43). Written per seed at trk_off[seed]-sv_base.
EG3D_HD uint32_t build_seed_view_map(const int32_t* views, uint32_t k, int32_t* map_view, uint32_t* map_entry) {
  uint32_t n = 0;
  for (uint32_t i = 0; i < k; i++) {
    int32_t v = views[i];
    uint32_t pos = 0;
    while (pos < n && map_view[pos] < v) pos++;
    if (pos < n && map_view[pos] == v) {
      map_entry[pos] = i;  // later duplicate overwrites
    } else {
      for (uint32_t j = n; j > pos; j--) {
        map_view[j] = map_view[j - 1];
        map_entry[j] = map_entry[j - 1];
      }
      map_view[pos] = v;
      map_entry[pos] = i;
      n++;
    }
  }
  return n;
}

struct TaskDesc {
  uint32_t seed, entry, hit;
  int32_t sel_view[3];
  uint32_t sel_entry[3];
  uint32_t cnt[3];
  uint32_t n_hyp;
};

// Choose the three views (min id, start view or median, max id of the non-empty lists).
EG3D_HD void task_setup(const StageAView& a, uint32_t t, const int32_t* map_view, const uint32_t* map_entry,
                        const uint32_t* map_n, TaskDesc& d) {
  d.seed = a.task_seed[t];
  d.entry = a.task_entry[t];
  d.hit = a.task_hit[t];
  d.n_hyp = 0;
  for (int k = 0; k < 3; k++) {
    d.sel_view[k] = -1;
    d.sel_entry[k] = 0;
    d.cnt[k] = 0;
  }
  const uint32_t base = track_base(a, d.seed);
  const uint32_t n = track_n_views(a, map_n, d.seed);
  const int32_t* mv = map_view + base;
  const uint32_t* me = map_entry + base;
  const uint32_t lo = a.task_list_off[t];
  const int32_t start_view = track_view(a, d.seed, d.entry);
  int non_empty = 0, min_j = -1, max_j = -1;
  for (uint32_t j = 0; j < n; j++)
    if (a.list_cnt[lo + me[j]] > 0) {
      non_empty++;
      if (min_j < 0) min_j = (int)j;
      max_j = (int)j;
    }
  if (non_empty < 3) return;
  int rel = 0, mid_j = 0;
  const int rel_mid = non_empty / 2;
  for (uint32_t j = 0; j < n; j++)
    if (a.list_cnt[lo + me[j]] > 0) {
      if (rel == rel_mid) {
        mid_j = (int)j;
        break;
      }
      rel++;
    }
  int sel_j[3];
  sel_j[0] = min_j;
  sel_j[2] = max_j;
  if (start_view == mv[min_j] || start_view == mv[max_j]) {
    sel_j[1] = mid_j;
  } else {
    int sj = 0;
    for (uint32_t j = 0; j < n; j++)
      if (mv[j] == start_view) sj = (int)j;
    sel_j[1] = sj;
  }
  uint64_t nh = 1;
  for (int k = 0; k < 3; k++) {
    d.sel_view[k] = mv[sel_j[k]];
    d.sel_entry[k] = me[sel_j[k]];
    d.cnt[k] = a.list_cnt[lo + d.sel_entry[k]];
    nh *= d.cnt[k];
  }
  d.n_hyp = nh > 0xfffffffeull ? 0xfffffffeu : (uint32_t)nh;
}

// hypothesis index -> its three hits (c0 slowest, c2 fastest: triangulation.cpp:565-567)
EG3D_HD void hypothesis_hits(const StageAView& a, const TaskDesc& d, uint32_t t, uint32_t local, Obs c[3]) {
  const uint32_t lo = a.task_list_off[t];
  uint32_t i2 = local % d.cnt[2];
  uint32_t r = local / d.cnt[2];
  uint32_t i1 = r % d.cnt[1];
  uint32_t i0 = r / d.cnt[1];
  c[0] = a.hits[a.list_ptr[lo + d.sel_entry[0]] + i0];
  c[1] = a.hits[a.list_ptr[lo + d.sel_entry[1]] + i1];
  c[2] = a.hits[a.list_ptr[lo + d.sel_entry[2]] + i2];
}

struct ChainSeed {
  uint32_t task;
  uint32_t winner;     // global hypothesis index
  uint32_t pts2_src;   // global hypothesis index whose direction-2 list is used, or 0xffffffff
  uint32_t n1, n2;
};

// Uniqueness rule: exactly one compatible hypothesis (Q3). The winner's direction-2 points
// come from the most recent earlier hypothesis of the task whose direction 2 was valid when
// its own is not (Q12).
EG3D_HD bool select_task(const HypResult* res, uint32_t h0, uint32_t h1, ChainSeed& cs) {
  uint32_t winner = 0xffffffffu;
  uint32_t last_d2 = 0xffffffffu, src = 0xffffffffu;
  for (uint32_t h = h0; h < h1; h++) {
    const uint32_t st = res[h].status;
    if (st & HYP_D2) last_d2 = h;
    if (st & HYP_COMPAT) {
      if (winner != 0xffffffffu) return false;
      winner = h;
      src = last_d2;  // == h when the winner's own direction 2 is valid
    }
  }
  if (winner == 0xffffffffu) return false;
  cs.winner = winner;
  cs.pts2_src = src;
  cs.n1 = res[winner].n1;
  cs.n2 = (src != 0xffffffffu) ? res[src].n2 : 0;
  return true;
}

// Scratch slice layout of one chain (bytes); all sub-arrays 16-byte aligned (Obs is one 128-bit word).
struct ChainLayout {
  uint32_t cap_pts, pool_cap, tmp_cap, n_views;
  size_t off_pts, off_pool, off_sdir, off_edir, off_p1, off_p2, off_cand, off_slots, off_epc, off_ta, off_tb, off_tm, off_mbox, total;
};
EG3D_HD size_t align8(size_t v) { return (v + 15) & ~(size_t)15; }  // (name kept: 16-byte alignment)
EG3D_HD ChainLayout chain_layout(uint32_t cap_pts, uint32_t pool_cap, uint32_t n_views) {
  ChainLayout L;
  L.cap_pts = cap_pts;
  L.pool_cap = pool_cap;
  L.n_views = n_views;
  L.tmp_cap = 2 * n_views + 8;
  // the small arrays first, the observation pool (of which a chain touches the beginning) last: what a chain keeps
  // warm is then one compact region at the start of its slice
  size_t o = 0;
  L.off_pts = o;
  o = align8(o + sizeof(ChainPt) * cap_pts);
  L.off_sdir = o;
  o = align8(o + sizeof(uint32_t) * n_views);
  L.off_edir = o;
  o = align8(o + sizeof(uint32_t) * n_views);
  L.off_p1 = o;
  o = align8(o + sizeof(Pending) * cap_pts);
  L.off_p2 = o;
  o = align8(o + sizeof(Pending) * cap_pts);
  L.off_cand = o;
  o = align8(o + sizeof(ViewCand) * cap_pts);
  L.off_slots = o;
  o = align8(o + sizeof(StepSlot) * EG3D_STEP_OBS);
  L.off_epc = o;
  o = align8(o + sizeof(EpcSolve) * cap_pts);
  L.off_ta = o;
  o = align8(o + sizeof(Obs) * L.tmp_cap);
  L.off_tb = o;
  o = align8(o + sizeof(Obs) * L.tmp_cap);
  L.off_tm = o;
  o = align8(o + L.tmp_cap);
  L.off_mbox = o;  // answer of a single solve of the chain state machine (eg3d_chain_sm.h: SmMbox, 16 bytes)
  o = align8(o + 16);
  L.off_pool = o;
  o = align8(o + sizeof(Obs) * pool_cap);
  L.total = o;
  return L;
}
EG3D_HD void chain_bind(Chain& c, const ChainLayout& L, unsigned char* slice) {
  c.pts = (ChainPt*)(slice + L.off_pts);
  c.cap_pts = (int32_t)L.cap_pts;
  c.pool = (Obs*)(slice + L.off_pool);
  c.pool_cap = L.pool_cap;
  c.start_dirs = (uint32_t*)(slice + L.off_sdir);
  c.end_dirs = (uint32_t*)(slice + L.off_edir);
  c.pend1 = (Pending*)(slice + L.off_p1);
  c.pend2 = (Pending*)(slice + L.off_p2);
  c.cand = (ViewCand*)(slice + L.off_cand);
  c.slots = (StepSlot*)(slice + L.off_slots);
  c.epcres = (EpcSolve*)(slice + L.off_epc);
  c.tmp_a = (Obs*)(slice + L.off_ta);
  c.tmp_b = (Obs*)(slice + L.off_tb);
  c.tmp_mask = (uint8_t*)(slice + L.off_tm);
  c.tmp_cap = (int32_t)L.tmp_cap;
}

struct ChainOut {
  uint32_t n_points, n_obs, flags, head;
  uint64_t bytes;
  uint64_t spt, sobs;  // where the finished chain was packed in the launch's staging area (k3b_expand)
  uint64_t tsec[16];  // diagnostic section ticks (zero unless built with EG3D_SECTION_TIMING)
};

// Build the chain reverse(pts1) + central + pts2, then offer it to every view except the
// three selected, ascending (triangulation.cpp:960-973). One lane, one chain.
// hyp_base = global index of the task's first hypothesis.
template <class Team>
EG3D_HD_FLAT void expand_chain(const Team& tm, const DevScene& s, const StageAView& a, const TaskDesc& d, const ChainSeed& cs,
                          uint32_t hyp_base, const HypResult* res, const HPoint* arena, const int32_t* map_view,
                          const uint32_t* map_entry, const uint32_t* map_n, const ChainLayout& L,
                          unsigned char* slice, ChainOut& out) {
  Chain c;
  chain_bind(c, L, slice);
  tm.bind(c);
  c.flags = 0;
  c.bytes = 0;
  for (int k = 0; k < 16; k++) c.tsec[k] = 0;
  const uint64_t t_begin = EG3D_TICK();
  c.pool_used = 0;
  const HypResult& w = res[cs.winner];
  const int L0 = (int)(cs.n1 + 1 + cs.n2);
  c.len = 0;
  c.head = ((int)L.cap_pts - L0) / 2;
  if (c.head < 0) {
    c.head = 0;
    c.flags |= 1u;
  }
  for (uint32_t v = 0; v < L.n_views; v++) {
    c.start_dirs[v] = 0;
    c.end_dirs[v] = 0;
  }
  for (int k = 0; k < 3; k++) {
    c.start_dirs[d.sel_view[k]] = w.dirs1[k];
    c.end_dirs[d.sel_view[k]] = w.dirs2[k];
  }
  // initial chain = direction-1 points reversed, the central point, direction-2 points
  // (new_3dpoint_and_sides_plgp_matches_to_vector, polyline_graph_2d.cpp:1298-1306). PARALLEL over
  // the points: every hypothesis-stage point has three observations => a block of 4 pool slots each.
  const int centre0 = (int)cs.n1;
  {
    int L1 = L0;
    if (c.head + L1 > c.cap_pts) {
      c.flags |= 1u;
      L1 = c.cap_pts - c.head;
    }
    if ((uint32_t)L1 * 4u > c.pool_cap) {
      c.flags |= 2u;
      L1 = (int)(c.pool_cap / 4u);
    }
    const uint32_t p2 = cs.pts2_src != 0xffffffffu ? res[cs.pts2_src].pts2_off : 0u;
    for (int i = tm.lane(); i < L1; i += tm.size()) {
      HPoint hp;
      if (i < centre0) {
        hp = arena[w.pts1_off + (uint32_t)(centre0 - 1 - i)];
      } else if (i == centre0) {
        hypothesis_hits(a, d, cs.task, cs.winner - hyp_base, hp.o);
        hp.X[0] = w.X[0];
        hp.X[1] = w.X[1];
        hp.X[2] = w.X[2];
        hp.nobs = 3;
        hp.pad = 0;
      } else {
        hp = arena[p2 + (uint32_t)(i - centre0 - 1)];
      }
      ChainPt p;
      p.X[0] = hp.X[0];
      p.X[1] = hp.X[1];
      p.X[2] = hp.X[2];
      p.off = 4u * (uint32_t)i;
      p.cap = 4;
      p.nobs = hp.nobs;
      for (uint32_t k = 0; k < hp.nobs; k++) c.pool[p.off + k] = hp.o[k];
      c.pts[c.head + i] = p;
    }
    c.len = L1;
    c.pool_used = 4u * (uint32_t)L1;
  }
  int centre = centre0;  // index of the central point; moves when the chain grows at the front
  tm.sync();
  EG3D_SEC_ADD(c.tsec, 9, EG3D_TICK() - t_begin);
  // every view except the three selected, ascending; epc = the task's hits in that view
  const uint32_t base = tm.uni(track_base(a, d.seed));
  const uint32_t n = tm.uni(track_n_views(a, map_n, d.seed));
  const int32_t* mv = map_view + base;
  const uint32_t* me = map_entry + base;
  const uint32_t lo = tm.uni(a.task_list_off[cs.task]);
  uint32_t j = 0;
  // (an initial chain that did not fit its slice is incomplete — its central point may be missing: nothing is expanded, the
  // capacity flag makes the host relaunch the chain with a larger slice)
  for (int v = 0; v < s.n_views && !(c.flags & 3u); v++) {
    if (v == d.sel_view[0] || v == d.sel_view[1] || v == d.sel_view[2]) continue;
    while (j < n && mv[j] < v) j++;
    const Obs* epc = nullptr;
    int n_epc = 0;
    if (j < n && mv[j] == v) {
      epc = a.hits + tm.uni(a.list_ptr[lo + me[j]]);
      n_epc = (int)tm.uni(a.list_cnt[lo + me[j]]);
    }
    const uint64_t tv0 = EG3D_TICK();
    expand_to_view(tm, s, c, v, epc, n_epc, centre);
    EG3D_SEC_ADD(c.tsec, 12, EG3D_TICK() - tv0);
  }
  uint32_t nobs = 0;
  for (int i = 0; i < c.len; i++) nobs += chain_at(c, i).nobs;
  out.n_points = (uint32_t)c.len;
  out.n_obs = nobs;
  out.flags = c.flags;
  out.head = (uint32_t)c.head;
  out.bytes = c.bytes;
  c.tsec[7] = EG3D_TICK() - t_begin;
  for (int k = 0; k < 16; k++) out.tsec[k] = c.tsec[k];
}

// A finished chain leaves its working slice as a packed record in the launch's STAGING area: n_points
// point headers (16 B) followed, in a second array, by the observations of those points back to back
// in chain order (16 B each). K4 turns the records into the ordered SoA output.
struct StagePt {
  float X[3];
  uint32_t nobs;
};
// Staging area of one K3b launch: bump-allocated by the chains as they finish (order of completion).
// The counters keep counting past the capacity, so the host learns the exact need of an overflowing
// launch and repeats it once with room for everything.
struct StageBuf {
  StagePt* pts;
  Obs* obs;
  unsigned long long cap_pts, cap_obs;
  unsigned long long* used;  // [0] points, [1] observations
};

// K4 body: copy one finished chain into the ordered SoA output.
EG3D_HD void emit_chain(const ChainLayout& L, const unsigned char* slice, const ChainOut& co, const TaskDesc& d,
                        uint64_t point_base, uint64_t obs_base, float* X, uint64_t* obs_off, int32_t* obs_view,
                        uint32_t* obs_pl, uint32_t* obs_seg, float* obs_xy, uint32_t* key) {
  const ChainPt* pts = (const ChainPt*)(slice + L.off_pts);
  const Obs* pool = (const Obs*)(slice + L.off_pool);
  uint64_t o = obs_base;
  for (uint32_t i = 0; i < co.n_points; i++) {
    const ChainPt& p = pts[co.head + i];
    const uint64_t pi = point_base + i;
    X[3 * pi] = p.X[0];
    X[3 * pi + 1] = p.X[1];
    X[3 * pi + 2] = p.X[2];
    obs_off[pi] = o;
    key[4 * pi] = d.seed;
    key[4 * pi + 1] = d.entry;
    key[4 * pi + 2] = d.hit;
    key[4 * pi + 3] = i;
    for (uint32_t k = 0; k < p.nobs; k++) {
      const Obs& po = pool[p.off + k];
      obs_view[o] = po.view;
      obs_pl[o] = po.pl;
      obs_seg[o] = po.seg;
      obs_xy[2 * o] = po.x;
      obs_xy[2 * o + 1] = po.y;
      o++;
    }
  }
}

}  // namespace eg3d
