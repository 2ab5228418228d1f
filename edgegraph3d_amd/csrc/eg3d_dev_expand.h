// eg3d_dev_expand.h — expand-all-views stage of the MI355X path: a chain of 3-D edge-points
// (found by a 3-view hypothesis) is offered to every other view in ascending view order; a
// view that sees the edge contributes one observation per chain point it can follow, each
// addition re-solving the point with FP64 Gauss-Newton, and may extend the chain at either end.
// One wavefront owns one chain (kernel k3b_expand).
//
// Behaviour reproduced (reference): expand_allpoints_to_other_view_using_plmap
// (src/edgegraph3d/utils/geometry/triangulation.cpp:742-833, SWITCH_DISABLE_INTERVAL branch),
// add_view_to_3dpoint_and_sides_plgp_matches_vector, compatible_direction_noupdate_vector,
// follow_direction_vector_start/_end, compatible() vector form
// (src/edgegraph3d/matching/plg_matching/plg_matching.cpp:1345-1412, 866-914, 771-795, 633-759).
// Design: the chain is a deque of fixed-size point headers in an HBM scratch slice with the
// observations of each point in a contiguous block of an append-only pool (blocks double when
// they fill); one wavefront owns one chain and spreads the Gauss-Newton solves over its lanes.
#pragma once
#include "eg3d_dev_follow.h"

namespace eg3d {

// A chain point: its observations are a CONTIGUOUS block pool[off .. off+nobs) of the chain's
// append-only pool (capacity `cap`; a full block is relocated to a block of twice the size at the
// pool's end), so a solve reads them with independent, coalescable loads — no pointer chasing —
// and a team can address observation k directly.
struct ChainPt {
  float X[3];
  uint32_t off, nobs, cap;
};
struct Pending {
  float X[3];
  Obs o;
  uint32_t ok;  // the ADD solve of this candidate succeeded
  uint32_t pad;
};
// Per chain point, for the view being offered: unique 4 px-grid polyline and closest point.
struct ViewCand {
  uint32_t valid;  // exactly one polyline in the window: 0x80000000 | its vertex count (0 = no candidate)
  uint32_t pl, seg;
  float x, y, d2;
  uint32_t cok;    // speculative central ADD solve of (point + this observation) succeeded
  float cX[3];     // its result
  uint32_t eok;    // epipolar line of the point's FIRST observation in the offered view is valid
  float ea, eb, ec;
};
// speculative central solve of one epipolar candidate against the chain's central point
struct EpcSolve {
  uint32_t ok;
  float X[3];
};

// One "starting observation" candidate of an N-view step, evaluated by one team member.
#define EG3D_STEP_OBS 16
struct StepSlot {
  uint32_t ok;     // candidate produced a valid point
  int32_t m;       // its observation count
  float X[3];
  uint32_t flags;  // EG3D_FLAG_* bits raised while evaluating it
  Obs sel[EG3D_STEP_OBS];
  Obs tmp[EG3D_STEP_OBS];
  uint8_t mask[EG3D_STEP_OBS];
};

// Per-chain working set (pointers into this chain's scratch slice).
struct Chain {
  ChainPt* pts;
  int32_t cap_pts;
  int32_t head;  // index of chain[0] in pts
  int32_t len;
  Obs* pool;
  uint32_t pool_used, pool_cap;
  uint32_t* start_dirs;  // [V] node id the chain front is heading to, per view
  uint32_t* end_dirs;    // [V]
  Pending* pend1;        // [cap_pts]
  Pending* pend2;        // [cap_pts]
  ViewCand* cand;        // [cap_pts]
  StepSlot* slots;       // [EG3D_STEP_OBS]
  EpcSolve* epcres;      // [cap_pts]
  Obs* tmp_a;            // [tmp_cap]
  Obs* tmp_b;            // [tmp_cap]
  uint8_t* tmp_mask;     // [tmp_cap]
  int32_t tmp_cap;
  uint32_t flags;
  uint64_t bytes;  // vertices of 4 px-grid polylines tested (algorithmic bytes)
  uint64_t tsec[16];  // diagnostic: shader-clock ticks per section (0 cand, 1 step walks, 2 side walks, 3 batch GN, 4 follow, 5 step DLT, 6 step GN, 7 whole, 8 commit, 9 chain init, 10 epc pre-solves, 11 new point, 12 expand_to_view, 13 attach_view, 14 3-subset fallback, 15 sequential N-view steps of the following)
};
#if defined(__HIP_DEVICE_COMPILE__) && defined(EG3D_SECTION_TIMING)
#define EG3D_TICK() ((uint64_t)__builtin_readcyclecounter())
#else
#define EG3D_TICK() ((uint64_t)0)
#endif
// -DEG3D_ONE_SECTION=k (with -DEG3D_SECTION_TIMING): the LIGHT timing build — only section k and the whole chain (7) are
// accumulated, and the Gauss-Newton counters are off: two live counters instead of sixteen, so the kernel keeps its
// registers and its speed (the full timing build runs ~10x slower and distorts the shares). tools/section_light.sh.
#ifdef EG3D_ONE_SECTION
#define EG3D_SEC_ON(i) ((i) == EG3D_ONE_SECTION || (i) == 7)
#else
#define EG3D_SEC_ON(i) true
#endif
#define EG3D_SEC_ADD(arr, i, v)            \
  do {                                     \
    if (EG3D_SEC_ON(i)) (arr)[i] += (v);   \
  } while (0)

EG3D_HD ChainPt& chain_at(Chain& c, int i) { return c.pts[c.head + i]; }

EG3D_HD void point_init(ChainPt& p) {
  p.off = 0;
  p.nobs = 0;
  p.cap = 0;
}
// Reserve room for `want` observations in a fresh block (used when a point is created).
EG3D_HD bool point_reserve(Chain& c, ChainPt& p, uint32_t want) {
  uint32_t cap = 4;
  while (cap < want) cap <<= 1;
  if (c.pool_used + cap > c.pool_cap) {
    c.flags |= 2u;
    return false;
  }
  p.off = c.pool_used;
  p.cap = cap;
  p.nobs = 0;
  c.pool_used += cap;
  return true;
}
EG3D_HD bool pool_append(Chain& c, ChainPt& p, const Obs& o) {
  if (p.nobs == p.cap) {
    const uint32_t ncap = p.cap ? p.cap * 2 : 4;
    if (c.pool_used + ncap > c.pool_cap) {
      c.flags |= 2u;
      return false;
    }
    const uint32_t noff = c.pool_used;
    for (uint32_t i = 0; i < p.nobs; i++) c.pool[noff + i] = c.pool[p.off + i];
    c.pool_used += ncap;
    p.off = noff;
    p.cap = ncap;
  }
  c.pool[p.off + p.nobs] = o;
  p.nobs++;
  return true;
}

// GN from the stored X with the point's observations plus one more (ADD).
EG3D_HD bool add_observation_solve(const DevScene& s, const Chain& c, const ChainPt& p, const Obs& extra,
                                   float Xout[3]) {
  double X0[3] = {(double)p.X[0], (double)p.X[1], (double)p.X[2]};
  const int n = (int)p.nobs;
  const Obs* a = c.pool + p.off;
  if (n + 1 <= EG3D_LOCAL_OBS) {
    // gather the block once (independent loads); the solver then streams from lane-private arrays
    LocalCursor lc;
    for (int i = 0; i < n; i++) {
      lc.v[i] = a[i].view;
      lc.x[i] = a[i].x;
      lc.y[i] = a[i].y;
    }
    lc.v[n] = extra.view;
    lc.x[n] = extra.x;
    lc.y[n] = extra.y;
    lc.n = n + 1;
    lc.i = 0;
    return gauss_newton_f64(s.cam_P, lc, X0, Xout);
  }
  ArrayCursor cur;
  cur.a = a;
  cur.n = n;
  cur.extra = &extra;
  cur.i = 0;
  return gauss_newton_f64(s.cam_P, cur, X0, Xout);
}

// Execution team of one chain. The expand stage is written SPMD-style: every member runs the
// same control flow on identical values ("uniform" sections, redundant across lanes) and only
// the marked parallel sections split items across members, communicating through the chain's
// scratch slice followed by sync(). TeamSeq (1 member) is the sequential semantics and what the
// host instantiation uses; the GPU kernel uses a 64-lane wavefront (TeamWave in
// eg3d_kernels.hip).
//
// Solver hooks: gn_array() is ONE Gauss-Newton solve inside a uniform section; add_solves() is a
// batch of B independent ADD solves, request j described by get(j, point, extra) -> wanted and
// answered through put(j, ok, X). The sequential team runs them one after the other; the
// wavefront team spreads the observations of the solves over its lanes (eg3d_dev_coopgn.h).
EG3D_HD bool lazy_presolve(const DevScene& s);  // defined below (central pre-solves)
struct TeamSeq {
  static constexpr bool kSlotStep = false;  // N-view step: plain sequential candidates
  static constexpr bool kSpecFollow = false; // chain following: one step at a time
  EG3D_HD int lane() const { return 0; }
  EG3D_HD int size() const { return 1; }
  EG3D_HD void sync() const {}
  EG3D_HD void bind(Chain&) const {}
  // rank of this member among the members whose flag is set (+ their number)
  EG3D_HD int rank(bool flag, int& total) const {
    total = flag ? 1 : 0;
    return 0;
  }
  EG3D_HD uint32_t or_reduce(uint32_t v) const { return v; }
  // the 2-view DLT of a uniform section (a wavefront team keeps the decomposition's matrices in LDS)
  EG3D_HD void dlt(const float* P1, float x1, float y1, const float* P2, float x2, float y2, double X0[3]) const {
    dlt2(P1, x1, y1, P2, x2, y2, X0);
  }
  // ... and the one of the rarely taken 3-subset fallback
  EG3D_HD void dlt_rare(const float* P1, float x1, float y1, const float* P2, float x2, float y2, double X0[3]) const {
    dlt2(P1, x1, y1, P2, x2, y2, X0);
  }
  // a value every member holds identically (a wavefront team keeps it in scalar registers)
  template <class T>
  EG3D_HD T uni(const T& v) const {
    return v;
  }
  // members that share one item of an n_items-wide parallel section (a power of two), and the
  // merge of their partial closest-point results (smaller distance, then smaller segment index)
  // number of leading j in [0, m) for which pred(j) holds
  template <class Pred>
  EG3D_HD int leading_true(int m, Pred pred) const {
    int k = 0;
    while (k < m && pred(k)) k++;
    return k;
  }
  // which of the candidate slots [base, base + 64) below `end` hold a candidate (bit k = slot base + k): the visit of a
  // view jumps from candidate to candidate instead of loading every point's entry to find most of them empty
  EG3D_HD uint64_t valid_mask(const Chain& c, int base, int end) const {
    uint64_t m = 0;
    for (int k = 0; k < 64 && base + k < end; k++)
      if (c.cand[base + k].valid) m |= 1ull << k;
    return m;
  }
  EG3D_HD int group_size(int) const { return 1; }
  EG3D_HD void group_best(int, float&, PlPt&) const {}
  // exclusive prefix sum of v over the members (+ the total)
  EG3D_HD uint32_t excl_scan(uint32_t v, uint32_t& total) const {
    total = v;
    return 0;
  }
  // candidates of one side walk (walk_side_candidates_core); a team may first stage the polyline
  // and the epipolar lines of the chain points ahead in fast memory
  // windowed (lazy) or all-at-once (eager) speculative central solves for this scene (a team built for a class of
  // scenes may know the answer at compile time)
  EG3D_HD bool lazy_presolve(const DevScene& s) const { return eg3d::lazy_presolve(s); }
  // walk_stage() is called once per attachment before its side walks (and again after a solver batch, which may have
  // overwritten what was staged); a team that keeps nothing staged ignores it
  EG3D_HD void walk_stage(const DevScene&, Chain&, int, const PlRef&, int, int, int) const {}
  EG3D_HD int side_walk(const DevScene& s, Chain& c, int view, const PlRef& pl, const Obs& from, uint32_t direction, int lo,
                        int ci, int hi, bool towards_start, Pending* out) const;
  EG3D_HD bool gn_array(const DevScene& s, const Obs* a, int n, const double X0[3], float Xout[3]) const {
    ArrayCursor cur;
    cur.a = a;
    cur.n = n;
    cur.extra = nullptr;
    cur.i = 0;
    return gauss_newton_f64(s.cam_P, cur, X0, Xout);
  }
  // one ADD solve of a plain observation array (the greedy phase of the 3-subset fallback): rows a[0..n) + extra, from X0
  EG3D_HD bool add_array(const DevScene& s, const Obs* a, int n, const Obs& extra, const float X0[3], float Xout[3]) const {
    ArrayCursor cur;
    cur.a = a;
    cur.n = n;
    cur.extra = &extra;
    cur.i = 0;
    const double X0d[3] = {(double)X0[0], (double)X0[1], (double)X0[2]};
    return gauss_newton_f64(s.cam_P, cur, X0d, Xout);
  }
  // one ADD solve inside a uniform section
  EG3D_HD bool add_one(const DevScene& s, const Chain& c, const ChainPt& p, const Obs& extra, float Xout[3]) const {
    return add_observation_solve(s, c, p, extra, Xout);
  }
  template <class Get, class Put>
  EG3D_HD void add_solves(const DevScene& s, Chain& c, int B, Get get, Put put) const {
    for (int j = 0; j < B; j++) {
      const ChainPt* pt = nullptr;
      Obs o;
      if (!get(j, pt, o)) continue;
      float X[3];
      const bool ok = add_observation_solve(s, c, *pt, o, X);
      put(j, ok, X);
    }
  }
};
// One member, but routed through the slot-based (parallel-capable) N-view step: lets the host
// instantiation exercise exactly the code path the 64-lane team runs.
struct TeamSeqSlots : TeamSeq {
  static constexpr bool kSlotStep = true;
};

// Walk phase of one candidate of the N-view step on chain point `cur`: observation `st`
// advances 10 px on its polyline (uniform), every other observation follows by a bounded
// (5..20 px) epipolar walk — PARALLEL over the observations, the survivors compacted in
// observation order. Returns the number of observations collected in sel (0 = candidate dead).
template <class Team>
EG3D_HD_FLAT int stepn_walks(const Team& tm, const DevScene& s, const Obs* co_all, int n, int st, const uint32_t* dirs,
                        Obs* sel, int sel_cap, uint32_t& flags) {
  const Obs so = co_all[st];
  PlRef ps = polyline_of(s, so.view, so.pl);
  PlPt p, q;
  p.seg = so.seg;
  p.x = so.x;
  p.y = so.y;
  uint32_t w = walk_by_distance(ps, p, dirs[so.view], EG3D_FOLLOW_STEP, q);
  if (w & WALK_BAD_DIR) flags |= 8u;
  if (w & WALK_EXTREME) return 0;
  int m = 0;
  sel[m].view = so.view;
  sel[m].pl = so.pl;
  sel[m].seg = q.seg;
  sel[m].x = q.x;
  sel[m].y = q.y;
  m++;
  uint32_t fl = 0;
  for (int i0 = 0; i0 < n; i0 += tm.size()) {
    const int i = i0 + tm.lane();
    bool found = false;
    Obs r;
    r.view = 0;
    r.pl = r.seg = 0;
    r.x = r.y = 0.f;
    if (i < n && i != st) {
      const Obs co = co_all[i];
      float la, lb, lc;
      if (epiline(s.F, s.F_valid, s.n_views, so.view, co.view, q.x, q.y, la, lb, lc)) {
        PlRef pk = polyline_of(s, co.view, co.pl);
        PlPt cp, rp;
        cp.seg = co.seg;
        cp.x = co.x;
        cp.y = co.y;
        uint32_t wr = walk_by_line(pk, cp, dirs[co.view], la, lb, lc, true, EG3D_FOLLOW_MIN, EG3D_FOLLOW_MAX, rp);
        if (wr & WALK_BAD_DIR) fl |= 8u;
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
    int total;
    const int rk = tm.rank(found, total);
    if (found && m + rk < sel_cap) sel[m + rk] = r;
    if (m + total > sel_cap) {
      flags |= 2u;
      m = sel_cap;
      break;
    }
    m += total;
  }
  flags |= tm.or_reduce(fl);
  tm.sync();
  return m < 3 ? 0 : m;
}

// TRI on an observation array inside a uniform section: DLT as triangulate_array, the
// Gauss-Newton solve through the team.
template <class Team>
EG3D_HD bool triangulate_array_team(const Team& tm, const DevScene& s, const Obs* a, int n, float Xout[3],
                                    uint32_t& flags, uint64_t* tsec = nullptr) {
  const uint64_t t0 = EG3D_TICK();
  int mi = 0;
  int32_t mv = a[0].view;
  for (int i = 0; i < n; i++)
    if (a[i].view < mv) {
      mv = a[i].view;
      mi = i;
    }
  const int la = n - 1;
  if (a[mi].view == a[la].view) flags |= 16u;
  double X0[3];
  tm.dlt(s.cam_P + (size_t)a[mi].view * 16, a[mi].x, a[mi].y, s.cam_P + (size_t)a[la].view * 16, a[la].x, a[la].y, X0);
  const uint64_t t1 = EG3D_TICK();
  const bool ok = tm.gn_array(s, a, n, X0, Xout);
  if (tsec) {
    EG3D_SEC_ADD(tsec, 5, t1 - t0);
    EG3D_SEC_ADD(tsec, 6, EG3D_TICK() - t1);
  }
  return ok;
}

// Triangulation fallback of a candidate whose all-observation solve failed: first valid 3-subset
// + greedy ADD (triangulation.cpp:1105-1158); compacts sel to the kept observations. Every solve goes through the TEAM
// (round 4): the 3-subset triangulations and the ADD solves of the greedy phase are the same requests the rest of the
// expand stage issues, so a wavefront team spreads their rows over its lanes instead of running a sequential solver
// redundantly on all 64 — same additions in the same order, same bits (the subsets are enumerated in the order
// prev_permutation visits {1,1,1,0,...}: ascending (i, j, k), k fastest).
template <class Team>
EG3D_HD int stepn_fallback(const Team& tm, const DevScene& s, Obs* sel, int m, Obs* tmp, uint8_t* mask, float Xout[3],
                           uint32_t& flags) {
  if (m <= 3) return 0;
  bool valid = false;
  int bi = 0, bj = 0, bk = 0;
  for (int i = 0; i < m - 2 && !valid; i++)
    for (int j = i + 1; j < m - 1 && !valid; j++)
      for (int k = j + 1; k < m && !valid; k++) {
        tmp[0] = sel[i];
        tmp[1] = sel[j];
        tmp[2] = sel[k];
        tm.sync();
        if (triangulate_array_team(tm, s, tmp, 3, Xout, flags)) {
          valid = true;
          bi = i;
          bj = j;
          bk = k;
        }
      }
  if (!valid) return 0;
  for (int i = 0; i < m; i++) mask[i] = (i == bi || i == bj || i == bk) ? 1 : 0;
  tmp[0] = sel[bi];
  tmp[1] = sel[bj];
  tmp[2] = sel[bk];
  int kept = 3;
  for (int i = 0; i < m; i++) {
    if (!mask[i]) {
      tm.sync();
      float Xn[3];
      if (tm.add_array(s, tmp, kept, sel[i], Xout, Xn)) {
        mask[i] = 1;
        Xout[0] = Xn[0];
        Xout[1] = Xn[1];
        Xout[2] = Xn[2];
        tmp[kept++] = sel[i];
      }
    }
  }
  tm.sync();
  int k = 0;
  for (int i = 0; i < m; i++)
    if (mask[i]) sel[k++] = sel[i];
  return k;
}

// N-view step (compatible() vector form, plg_matching.cpp:633-759): the first candidate, in
// observation order, that yields a valid point wins. Team version: (1) PARALLEL walk phase, one
// member per starting observation; (2) PARALLEL, convergent all-observation triangulation of
// every live candidate; (3) uniform scan in observation order — a candidate whose solve failed
// runs the (rare, expensive) 3-subset fallback only when the sequential order actually reaches
// it. Flags of candidates past the winner are dropped, as the sequential order never ran them.
// The new point is returned in tmp_a[0..m) with its X.
// st0 = first starting observation to try (the sequential semantics start at 0; a team that has already evaluated the
// candidates before st0 itself — the look-ahead rounds of the wavefront team, redoing a step — passes where to go on).
template <class Team>
EG3D_HD_FLAT int stepn_chain(const Team& tm, const DevScene& s, Chain& c, const ChainPt& cur, const uint32_t* dirs,
                        float Xout[3], int st0 = 0) {
  const int n = (int)cur.nobs;
  if (n > EG3D_STEP_OBS || !Team::kSlotStep) {
    for (int st = st0; st < n; st++) {
      const uint64_t tw0 = EG3D_TICK();
      int m = stepn_walks(tm, s, c.pool + cur.off, n, st, dirs, c.tmp_a, c.tmp_cap, c.flags);
      EG3D_SEC_ADD(c.tsec, 1, EG3D_TICK() - tw0);
      if (!m) continue;
      if (triangulate_array_team(tm, s, c.tmp_a, m, Xout, c.flags, c.tsec)) return m;
      const uint64_t tfb0 = EG3D_TICK();
      m = stepn_fallback(tm, s, c.tmp_a, m, c.tmp_b, c.tmp_mask, Xout, c.flags);
      EG3D_SEC_ADD(c.tsec, 14, EG3D_TICK() - tfb0);
      if (m) return m;
    }
    return 0;
  }
  tm.sync();
  for (int st = tm.lane(); st < n; st += tm.size()) {
    StepSlot& sl = c.slots[st];
    uint32_t fl = 0;
    sl.m = stepn_walks(TeamSeq(), s, c.pool + cur.off, n, st, dirs, sl.sel, EG3D_STEP_OBS, fl);
    sl.flags = fl;
    sl.ok = 0;
  }
  tm.sync();
  for (int st = tm.lane(); st < n; st += tm.size()) {
    StepSlot& sl = c.slots[st];
    if (sl.m) {
      uint32_t fl = sl.flags;
      float X[3] = {0.f, 0.f, 0.f};
      sl.ok = triangulate_array(s.cam_P, sl.sel, sl.m, X, fl) ? 1u : 0u;
      sl.X[0] = X[0];
      sl.X[1] = X[1];
      sl.X[2] = X[2];
      sl.flags = fl;
    }
  }
  tm.sync();
  for (int st = 0; st < n; st++) {
    StepSlot& sl = c.slots[st];
    c.flags |= sl.flags;
    int m = sl.m;
    if (!m) continue;
    if (sl.ok) {
      Xout[0] = sl.X[0];
      Xout[1] = sl.X[1];
      Xout[2] = sl.X[2];
    } else {
      for (int i = 0; i < m; i++) c.tmp_a[i] = sl.sel[i];
      const uint64_t tfb0 = EG3D_TICK();
      m = stepn_fallback(tm, s, c.tmp_a, m, c.tmp_b, c.tmp_mask, Xout, c.flags);
      EG3D_SEC_ADD(c.tsec, 14, EG3D_TICK() - tfb0);
      if (!m) continue;
      return m;
    }
    for (int i = 0; i < m; i++) c.tmp_a[i] = sl.sel[i];
    return m;
  }
  return 0;
}

EG3D_HD bool new_point_from_list(Chain& c, ChainPt& np, const Obs* list, int m, const float X[3]) {
  np.X[0] = X[0];
  np.X[1] = X[1];
  np.X[2] = X[2];
  point_init(np);
  if (!point_reserve(c, np, (uint32_t)m + 1)) return false;
  for (int i = 0; i < m; i++)
    if (!pool_append(c, np, list[i])) return false;
  return true;
}
EG3D_HD bool new_point_from_tmp(Chain& c, ChainPt& np, int m, const float X[3]) {
  np.X[0] = X[0];
  np.X[1] = X[1];
  np.X[2] = X[2];
  point_init(np);
  if (!point_reserve(c, np, (uint32_t)m + 1)) return false;
  for (int i = 0; i < m; i++)
    if (!pool_append(c, np, c.tmp_a[i])) return false;
  return true;
}

// Grow the chain at its front (towards start_dirs) or back (towards end_dirs) while N-view steps
// succeed (follow_direction_vector_start / _end, plg_matching.cpp:771-795). Returns points added.
template <class Team>
EG3D_HD_FLAT int follow_end(const Team& tm, const DevScene& s, Chain& c, bool front) {
  if constexpr (Team::kSpecFollow) return tm.follow(s, c, front);
  int added = 0;
  for (;;) {
    float X[3];
    int m = stepn_chain(tm, s, c, front ? chain_at(c, 0) : chain_at(c, c.len - 1), front ? c.start_dirs : c.end_dirs, X);
    if (m == 0) break;
    if (front ? (c.head <= 0) : (c.head + c.len >= c.cap_pts)) {
      c.flags |= 1u;
      break;
    }
    ChainPt np;
    const uint64_t tn0 = EG3D_TICK();
    if (!new_point_from_tmp(c, np, m, X)) break;
    if (front) {
      c.head--;
      c.pts[c.head] = np;
    } else {
      c.pts[c.head + c.len] = np;
    }
    EG3D_SEC_ADD(c.tsec, 11, EG3D_TICK() - tn0);
    c.len++;
    added++;
  }
  return added;
}

// Side walk, phase 1 (uniform): from `from` on view `view` towards node `direction`, visit chain
// points ci-1, ci-2, ... >= lo (towards_start) or ci+1, ... < hi; for each take the next hit of
// the epipolar line of the point's FIRST observation. Walk positions do not depend on the
// solver, so all candidates are generated first; returns how many walks succeeded.
// walk(pl, from, direction, a, b, c, out) = the unbounded next-hit walk (walk_by_line, or a team's
// segment-parallel version of it).
template <class PlT, class EpiPtr, class WalkFn>
EG3D_HD int walk_side_candidates_core(const DevScene& s, Chain& c, const PlT& pl, EpiPtr epi, int n_epi, int view,
                                      const Obs& from, uint32_t direction, int lo, int ci, int hi, bool towards_start,
                                      Pending* out, WalkFn walk) {
  int cnt = 0;
  int t = 0;
  PlPt actual;
  actual.seg = from.seg;
  actual.x = from.x;
  actual.y = from.y;
  int i = towards_start ? ci - 1 : ci + 1;
  while ((towards_start && i >= lo) || (!towards_start && i < hi)) {
    // epipolar line of chain point i's first observation in `view`: precomputed lane-parallel by
    // view_epilines() for the view being offered
    float la, lb, lc;
    if (t < n_epi) {
      if (epi[4 * t] == 0.0f) break;
      la = epi[4 * t + 1];
      lb = epi[4 * t + 2];
      lc = epi[4 * t + 3];
    } else {
      const ViewCand& ve = c.cand[c.head + i];
      if (!ve.eok) break;
      la = ve.ea;
      lb = ve.eb;
      lc = ve.ec;
    }
    t++;
    PlPt nx;
    uint32_t w = walk(pl, actual, direction, la, lb, lc, nx);
    if (w & WALK_BAD_DIR) c.flags |= 8u;
    if (!(w & WALK_FOUND)) break;
    Pending& pd = out[cnt++];
    pd.o.view = view;
    pd.o.pl = from.pl;
    pd.o.seg = nx.seg;
    pd.o.x = nx.x;
    pd.o.y = nx.y;
    pd.ok = 0;
    actual = nx;
    if (towards_start)
      i--;
    else
      i++;
  }
  return cnt;
}
EG3D_HD int TeamSeq::side_walk(const DevScene& s, Chain& c, int view, const PlRef& pl, const Obs& from, uint32_t direction,
                               int lo, int ci, int hi, bool towards_start, Pending* out) const {
  return walk_side_candidates_core(s, c, pl, (const float*)nullptr, 0, view, from, direction, lo, ci, hi,
                                   towards_start, out,
                                   [](const PlRef& p, const PlPt& a, uint32_t d, float la, float lb, float lc, PlPt& nx) {
                                     return walk_by_line(p, a, d, la, lb, lc, false, 0.0f, 0.0f, nx);
                                   });
}

// The side walks of an attachment and their ADD solves. One ORIENTATION = the walk towards dS on the chain's start side
// and towards dE on its end side; candidate j of the start side goes against chain point ci-1-j, of the end side
// against ci+1+j; n1 / n2 = leading successes of each side — the reference stops a side at its first failed walk or
// solve (plg_matching.cpp:866-914) and walks the end side only when the start side attached something (:1345-1412).
// The end-side walk is made before the start side's solves are known (its positions do not depend on them) whenever
// the start side produced candidates at all; it is discarded if none of those survives its solve. The reference tries
// the orientation (start, end) first and (end, start) when that one attaches nothing on the start side. Walks do not
// depend on solves, so: stage once (tm.walk_stage), walk orientation A; if its start side finds NOTHING, orientation
// B is walked at once from the same staged data (no solver call in between); one batch of solves; only when A found
// candidates whose first solve failed is B walked and solved on its own (after re-staging: the solver's staging
// area aliases the walk's). Same walks, same solves that count, same flags as two independent passes.
// Returns the orientation that attached (0 = none, 1 = A, 2 = B) with n1 / n2.
template <class Team>
EG3D_HD_FLAT int walk_sides_both(const Team& tm, const DevScene& s, Chain& c, int view, const Obs& from, const PlRef& pl, int lo,
                                 int ci, int hi, int& n1, int& n2) {
  Pending* p1 = c.pend1;
  Pending* p2 = c.pend2;
  auto walks = [&](uint32_t dS, uint32_t dE, int& m1, int& m2) {
    m1 = tm.side_walk(s, c, view, pl, from, dS, lo, ci, hi, true, p1);
    m2 = 0;
    if (m1 > 0 && ci < hi) m2 = tm.side_walk(s, c, view, pl, from, dE, lo, ci, hi, false, p2);
  };
  auto solves = [&](int m1, int m2) {
    tm.sync();
    const uint64_t t1 = EG3D_TICK();
    tm.add_solves(
        s, c, m1 + m2,
        [&](int j, const ChainPt*& pt, Obs& o) {
          if (j < m1) {
            pt = &chain_at(c, ci - 1 - j);
            o = p1[j].o;
          } else {
            pt = &chain_at(c, ci + 1 + (j - m1));
            o = p2[j - m1].o;
          }
          return true;
        },
        [&](int j, bool ok, const float* X) {
          if (!ok) return;
          Pending& pd = j < m1 ? p1[j] : p2[j - m1];
          pd.X[0] = X[0];
          pd.X[1] = X[1];
          pd.X[2] = X[2];
          pd.ok = 1;
        });
    tm.sync();
    n1 = tm.leading_true(m1, [&](int j) { return p1[j].ok != 0; });
    n2 = n1 > 0 ? tm.leading_true(m2, [&](int j) { return p2[j].ok != 0; }) : 0;
    EG3D_SEC_ADD(c.tsec, 3, EG3D_TICK() - t1);
  };
  n1 = n2 = 0;
  uint64_t t0 = EG3D_TICK();
  tm.walk_stage(s, c, view, pl, lo, ci, hi);
  int m1, m2;
  walks(pl.start, pl.end, m1, m2);
  int which = 1;
  if (m1 == 0) {  // orientation A cannot reach the lower neighbour: B, from the same staged data
    which = 2;
    walks(pl.end, pl.start, m1, m2);
  }
  EG3D_SEC_ADD(c.tsec, 2, EG3D_TICK() - t0);
  if (m1 == 0) return 0;
  solves(m1, m2);
  if (n1 > 0) return which;
  if (which == 2) return 0;
  // A's first start-side solve failed: orientation B on its own
  t0 = EG3D_TICK();
  tm.walk_stage(s, c, view, pl, lo, ci, hi);
  walks(pl.end, pl.start, m1, m2);
  EG3D_SEC_ADD(c.tsec, 2, EG3D_TICK() - t0);
  if (m1 == 0) return 0;
  solves(m1, m2);
  return n1 > 0 ? 2 : 0;
}

// Try to attach observation `o` of view o.view to chain point ci, then to its neighbours within
// [lo, hi). On success returns true with (to_start, to_end) = observations added on each side
// including newly grown points. (add_view_to_3dpoint_and_sides_plgp_matches_vector, Q13.)
template <class Team>
EG3D_HD_FLAT bool attach_view(const Team& tm, const DevScene& s, Chain& c, const Obs& o, int lo, int ci, int hi,
                         int& to_start, int& to_end, const uint32_t* pre_ok, const float* pre_X) {
  to_start = 0;
  to_end = 0;
  float Xc[3];
  if (pre_ok) {
    // the central solve was done speculatively (chain state unchanged since): reuse it
    if (!tm.uni(*pre_ok)) return false;
    Xc[0] = tm.uni(pre_X[0]);
    Xc[1] = tm.uni(pre_X[1]);
    Xc[2] = tm.uni(pre_X[2]);
  } else {
    uint64_t t0 = EG3D_TICK();
    bool okc = tm.add_one(s, c, chain_at(c, ci), o, Xc);
    EG3D_SEC_ADD(c.tsec, 1, EG3D_TICK() - t0);
    if (!okc) return false;
  }
  const int view = o.view;
  PlRef pl = polyline_of(s, view, o.pl);
  uint32_t nd1 = 0, nd2 = 0;
  int n1 = 0, n2 = 0;
  if (ci > lo) {
    const int which = walk_sides_both(tm, s, c, view, o, pl, lo, ci, hi, n1, n2);
    if (which == 1) {
      nd1 = pl.start;
      nd2 = pl.end;
    } else if (which == 2) {
      nd1 = pl.end;
      nd2 = pl.start;
    }
    // else: neither orientation reaches the lower neighbour (n1 = n2 = 0); with ci > lo >= 0 the
    // attachment is rejected below whatever the upper side would give.
  }
  if (ci > 0 && n1 == 0) return false;
  if (ci < c.len - 1 && n2 == 0) return false;
  // commit — PARALLEL over the 1 + n1 + n2 touched points (distinct points; blocks that are full
  // are relocated to space reserved by a prefix sum over the members)
  const uint64_t tcm0 = EG3D_TICK();
  {
    const int T = 1 + n1 + n2;
    for (int t0 = 0; t0 < T; t0 += tm.size()) {
      const int t = t0 + tm.lane();
      const bool act = t < T;
      ChainPt* p = nullptr;
      const Pending* pd = nullptr;
      if (act) {
        if (t == 0)
          p = &chain_at(c, ci);
        else if (t <= n1) {
          p = &chain_at(c, ci - t);
          pd = &c.pend1[t - 1];
        } else {
          p = &chain_at(c, ci + (t - n1));
          pd = &c.pend2[t - n1 - 1];
        }
      }
      uint32_t nobs = 0, cap = 0, off = 0, need = 0;
      if (act) {
        nobs = p->nobs;
        cap = p->cap;
        off = p->off;
        if (nobs == cap) need = cap ? cap * 2 : 4;
      }
      uint32_t total;
      const uint32_t ex = tm.excl_scan(need, total);
      if (c.pool_used + total > c.pool_cap) {
        c.flags |= 2u;  // the host enlarges the pool and reruns the chunk
        break;
      }
      if (act) {
        if (need) {
          const uint32_t noff = c.pool_used + ex;
          for (uint32_t i = 0; i < nobs; i++) c.pool[noff + i] = c.pool[off + i];
          off = noff;
          p->off = noff;
          p->cap = need;
        }
        if (pd) {
          c.pool[off + nobs] = pd->o;
          p->X[0] = pd->X[0];
          p->X[1] = pd->X[1];
          p->X[2] = pd->X[2];
        } else {
          c.pool[off + nobs] = o;
          p->X[0] = Xc[0];
          p->X[1] = Xc[1];
          p->X[2] = Xc[2];
        }
        p->nobs = nobs + 1;
      }
      c.pool_used += total;
    }
    tm.sync();
  }
  to_start = n1;
  to_end = n2;
  uint64_t tf0 = EG3D_TICK();
  EG3D_SEC_ADD(c.tsec, 8, tf0 - tcm0);
  // grow the chain at the front, then at the back (one call site: the following code is large)
  for (int side = 0; side < 2; side++) {
    const bool front = side == 0;
    if (front ? !(n1 > 0 && n1 == ci) : !(n2 > 0 && n2 == (c.len - ci - 1))) continue;
    if (front)
      c.start_dirs[view] = nd1;
    else
      c.end_dirs[view] = nd2;
    const int g = follow_end(tm, s, c, front);
    if (front) {
      to_start += g;
      ci += g;
    } else {
      to_end += g;
    }
  }
  EG3D_SEC_ADD(c.tsec, 4, EG3D_TICK() - tf0);
  tm.sync();
  return true;
}

// Unique polyline in the 3x3 (shrunk on boundaries) window of the 4 px grid around (x,y).
EG3D_HD bool unique_polyline_4px(const DevScene& s, int view, float x, float y, uint32_t& pl_id) {
  CellWindow w = cell_window(4.0f, s.width, s.height, s.g4_w, s.g4_h, x, y);
  if (w.c1 < w.c0) return false;
  bool have = false;
  uint32_t first = 0;
  const size_t base = (size_t)view * (size_t)(s.g4_w * s.g4_h);
  for (int r = w.r0; r <= w.r1; r++) {
    // cells of one row are contiguous in the CSR: scan [off(r,c0), off(r,c1+1))
    uint32_t a = s.g4_off[base + (size_t)r * s.g4_w + w.c0];
    uint32_t b = s.g4_off[base + (size_t)r * s.g4_w + w.c1 + 1];
    for (uint32_t k = a; k < b; k++) {
      uint32_t id = s.g4_ids[k];
      if (!have) {
        have = true;
        first = id;
      } else if (id != first)
        return false;
    }
  }
  pl_id = first;
  return have;
}

// The speculative central ADD solves of a view's candidates: all chain points at once after the
// candidate pass ("eager": one big batch, ~1/4 of the solves end up used), or a window of points at a
// time as the visit reaches them ("lazy"). Eager wins where the solves are short (C2: 8 views), lazy
// where a wasted solve is expensive: from 16 views on, in windows of 16 points (C3', 25 views: K3b
// 54.7 -> 53.6-53.9 ms), of 8 points from 64 views on (C4, 200 views: K3b -9 % against eager; windows
// of 16 / 32: +1.7 / +3.7 %). Chosen per scene by the number of views.
#ifndef EG3D_LAZY_PRESOLVE_MIN_VIEWS
#define EG3D_LAZY_PRESOLVE_MIN_VIEWS 16
#endif
EG3D_HD bool lazy_presolve(const DevScene& s) { return s.n_views >= EG3D_LAZY_PRESOLVE_MIN_VIEWS; }
#ifndef EG3D_PRESOLVE_WINDOW
#define EG3D_PRESOLVE_WINDOW 16      /* chain points per window of speculative central solves, scenes of 16..63 views */
#define EG3D_PRESOLVE_WINDOW_MANY 2  /*   ... of 64 views and more (round 5: 8 / 4 / 2 -> C4 step 1854 / 1792 / 1788 ms; 16 / 32 were +1.7 / +3.7 % on 8) */
#endif
EG3D_HD int presolve_window(const DevScene& s) { return s.n_views >= 64 ? EG3D_PRESOLVE_WINDOW_MANY : EG3D_PRESOLVE_WINDOW; }
// speculative central ADD solve of the chain points [from, to) whose candidate is within 4 px
template <class Team>
EG3D_HD_FLAT void central_presolves(const Team& tm, const DevScene& s, Chain& c, int v, int from, int to) {
  tm.add_solves(
      s, c, to - from,
      [&](int j, const ChainPt*& pt, Obs& o) {
        const ViewCand& vc = c.cand[c.head + from + j];
        if (!vc.valid || !(vc.d2 <= 16.0f)) return false;
        pt = &chain_at(c, from + j);
        o.view = v;
        o.pl = vc.pl;
        o.seg = vc.seg;
        o.x = vc.x;
        o.y = vc.y;
        return true;
      },
      [&](int j, bool ok, const float* X) {
        ViewCand& vc = c.cand[c.head + from + j];
        vc.cok = ok ? 1u : 0u;
        vc.cX[0] = X[0];
        vc.cX[1] = X[1];
        vc.cX[2] = X[2];
      });
  tm.sync();
}

// PARALLEL over chain points [from, len): project the point into view v, look up the unique
// polyline of the 4 px grid and its closest point (triangulation.cpp:791-806).
template <class Team>
EG3D_HD_FLAT void view_candidates(const Team& tm, const DevScene& s, Chain& c, int v, int from) {
  const float* P = s.cam_P + (size_t)v * 16;
  uint64_t tc0 = EG3D_TICK();
  // G members share a chain point when the chain is short: each scans 1/G of the polyline's segments
  const int n_items = c.len - from;
  const int G = tm.group_size(n_items);
  const int per_pass = tm.size() / G;
  for (int i0 = 0; i0 < n_items; i0 += per_pass) {
    const int item = i0 + tm.lane() / G, part = tm.lane() % G;
    const int i = from + item;
    const bool act = item < n_items;
    ViewCand vc;
    vc.valid = 0;
    vc.pl = 0;
    vc.seg = 0;
    vc.x = vc.y = vc.d2 = 0.0f;
    vc.cok = 0;
    vc.cX[0] = vc.cX[1] = vc.cX[2] = 0.0f;
    vc.eok = 0;
    vc.ea = vc.eb = vc.ec = 0.0f;
    float d2 = __builtin_inff();
    PlPt cp;
    cp.seg = 0xffffffffu;
    cp.x = cp.y = 0.0f;
    bool have = false;
    uint32_t nvtx = 0;
    if (act) {
      const ChainPt& pt = chain_at(c, i);
      if (part == 0) {
        const Obs& first = c.pool[pt.off];
        vc.eok = epiline(s.F, s.F_valid, s.n_views, first.view, v, first.x, first.y, vc.ea, vc.eb, vc.ec) ? 1u : 0u;
      }
      float u, w;
      project_f32(P, pt.X[0], pt.X[1], pt.X[2], u, w);
      uint32_t pl_id;
      if (unique_polyline_4px(s, v, u, w, pl_id)) {
        PlRef pl = polyline_of_bb(s, v, pl_id);
        const uint32_t nseg = pl.n - 1u;
        const uint32_t chunk = (nseg + (uint32_t)G - 1u) / (uint32_t)G;
        const uint32_t s0 = (uint32_t)part * chunk;
        const uint32_t s1 = s0 + chunk < nseg ? s0 + chunk : nseg;
        if (s0 < s1 || part == 0) d2 = polyline_closest_pruned(pl, u, w, s0 < nseg ? s0 : nseg, s1, cp);
        have = true;
        nvtx = pl.n;
        vc.pl = pl_id;
      }
    }
    tm.group_best(G, d2, cp);
    if (act && part == 0) {
      if (have) {
        vc.valid = 0x80000000u | nvtx;
        vc.d2 = d2;
        vc.seg = cp.seg;
        vc.x = cp.x;
        vc.y = cp.y;
      }
      c.cand[c.head + i] = vc;
    }
  }
  tm.sync();
  EG3D_SEC_ADD(c.tsec, 0, EG3D_TICK() - tc0);
  if (!tm.lazy_presolve(s)) {
    const uint64_t tp0 = EG3D_TICK();
    central_presolves(tm, s, c, v, from, c.len);
    EG3D_SEC_ADD(c.tsec, 10, EG3D_TICK() - tp0);  // (diagnostic: all speculative central solves are booked with the epc pre-solves)
  }
}

// Offer the chain to view v. epc = the task's epipolar hits in v (may be empty).
template <class Team>
EG3D_HD_FLAT void expand_to_view(const Team& tm, const DevScene& s, Chain& c, int v, const Obs* epc, int n_epc,
                            int& centre) {
  bool epc_matched = false;
  int idx_first = 0, idx_second = 0;
  // PARALLEL over the epipolar candidates: speculative central solves against chain[centre]
  // (results parked in the candidate array, which is rebuilt below before its own use)
  const int n_pre = n_epc < c.cap_pts ? n_epc : c.cap_pts;
  const uint64_t te0 = EG3D_TICK();
  if (n_epc > 0) {
    // PARALLEL: epipolar lines of every chain point in view v (the side walks read them)
    for (int i = tm.lane(); i < c.len; i += tm.size()) {
      const Obs& first = c.pool[chain_at(c, i).off];
      ViewCand& vc = c.cand[c.head + i];
      vc.eok = epiline(s.F, s.F_valid, s.n_views, first.view, v, first.x, first.y, vc.ea, vc.eb, vc.ec) ? 1u : 0u;
    }
    tm.add_solves(
        s, c, n_pre,
        [&](int e, const ChainPt*& pt, Obs& o) {
          pt = &chain_at(c, centre);
          o = epc[e];
          return true;
        },
        [&](int e, bool ok, const float* X) {
          EpcSolve r;
          r.ok = ok ? 1u : 0u;
          r.X[0] = X[0];
          r.X[1] = X[1];
          r.X[2] = X[2];
          c.epcres[e] = r;
        });
    tm.sync();
  }
  EG3D_SEC_ADD(c.tsec, 10, EG3D_TICK() - te0);
  for (int e = 0; e < n_epc; e++) {
    int a, b;
    const bool pre = e < n_pre;
    const uint64_t ta0 = EG3D_TICK();
    const bool att = attach_view(tm, s, c, epc[e], 0, centre, c.len, a, b, pre ? &c.epcres[e].ok : nullptr,
                                 pre ? c.epcres[e].X : nullptr);
    EG3D_SEC_ADD(c.tsec, 13, EG3D_TICK() - ta0);
    if (att) {
      epc_matched = true;
      if (a > centre) {
        centre = a;
        idx_first = 0;
        idx_second = a + b;
      } else {
        idx_first = centre - a;
        idx_second = centre + b;
      }
      break;
    }
  }
  int last_matched = -1;
  view_candidates(tm, s, c, v, 0);
  int spec_slot_hi = 0;  // lazy mode: pre-solves exist for the visited slots below this one
  // The visit goes from candidate to candidate (round 6): a point without one is a no-op of the loop, and most points of
  // most views have none — loading each point's entry to find that out was a dependent trip to memory per point and
  // view. The candidate array is indexed by absolute slot and was filled for the slots [head, head + len) of this
  // moment; `valid` never changes afterwards, and the points a following appends are never visited in this view.
  const int slot_hi0 = c.head + c.len;
  int mask_base = 0;
  uint64_t vmask = 0;
  bool have_mask = false;
  for (int cur = 0; cur < c.len; cur++) {
    {
      int slot = c.head + cur;
      while (slot < slot_hi0) {
        if (!have_mask || slot < mask_base || slot >= mask_base + 64) {
          mask_base = slot;
          vmask = tm.valid_mask(c, mask_base, slot_hi0);
          have_mask = true;
        }
        const uint64_t m = vmask >> (slot - mask_base);
        if (m) {
          slot += __builtin_ctzll(m);
          break;
        }
        slot = mask_base + 64;
      }
      int nxt = slot < slot_hi0 ? slot - c.head : c.len;  // the next point with a candidate (c.len: none)
      if (epc_matched && cur <= idx_first && idx_first < nxt) nxt = idx_first;  // ... or where the epipolar attachment's points start
      if (nxt >= c.len) break;
      cur = nxt;
    }
    if (epc_matched && cur == idx_first) {
      cur = idx_second;
      last_matched = idx_second;
      continue;
    }
    const ViewCand vc = tm.uni(c.cand[c.head + cur]);
    if (!vc.valid) continue;
    c.bytes += 8ull * (vc.valid & 0x7fffffffu);  // the polyline's vertex count (view_candidates)
    if (vc.d2 > 16.0f) return;  // abandons this view (Q4)
    if (tm.lazy_presolve(s) && c.head + cur >= spec_slot_hi) {
      int to = cur + presolve_window(s);
      if (to > c.len) to = c.len;
      const uint64_t tp0 = EG3D_TICK();
      central_presolves(tm, s, c, v, cur, to);
      EG3D_SEC_ADD(c.tsec, 10, EG3D_TICK() - tp0);
      spec_slot_hi = c.head + to;
    }
    Obs o;
    o.view = v;
    o.pl = vc.pl;
    o.seg = vc.seg;
    o.x = vc.x;
    o.y = vc.y;
    int hi = epc_matched ? (cur <= idx_first ? idx_first : c.len) : c.len;
    int a, b;
    const uint64_t ta0 = EG3D_TICK();
    const bool att = attach_view(tm, s, c, o, last_matched + 1, cur, hi, a, b, &c.cand[c.head + cur].cok, c.cand[c.head + cur].cX);
    EG3D_SEC_ADD(c.tsec, 13, EG3D_TICK() - ta0);
    if (att) {
      if (a > cur) {
        centre = a;
        cur = a + b;
      } else {
        cur = cur + b;
      }
      last_matched = cur;
      // the points still to visit (> cur) were not touched by the attachment and the candidate
      // array is indexed by absolute slot, so their candidates stay valid even if the chain grew
      // at the front; points appended at the back are never visited in this view (cur = len-1)
    }
  }
}

}  // namespace eg3d
