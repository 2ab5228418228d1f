// eg3d_dev_expand.h — expand-all-views stage of the MI355X path: a chain of 3-D edge-points
// (found by a 3-view hypothesis) is offered to every other view in ascending view order; a
// view that sees the edge contributes one observation per chain point it can follow, each
// addition re-solving the point with FP64 Gauss-Newton, and may extend the chain at either end.
// One GPU lane owns one chain (kernel k3b_expand).
//
// Behaviour reproduced (reference): expand_allpoints_to_other_view_using_plmap
// (src/edgegraph3d/utils/geometry/triangulation.cpp:742-833, SWITCH_DISABLE_INTERVAL branch),
// add_view_to_3dpoint_and_sides_plgp_matches_vector, compatible_direction_noupdate_vector,
// follow_direction_vector_start/_end, compatible() vector form
// (src/edgegraph3d/matching/plg_matching/plg_matching.cpp:1345-1412, 866-914, 771-795, 633-759).
// Design: the chain is a deque of fixed-size point headers in an HBM scratch slice with the
// observations of each point in an append-only pool threaded as singly linked lists, so that
// adding a view never moves data; all solver state stays in registers.
#pragma once
#include "eg3d_dev_follow.h"

namespace eg3d {

struct PoolObs {
  Obs o;
  uint32_t next;  // index in pool, 0xffffffff = end
};
struct ChainPt {
  float X[3];
  uint32_t head, tail, nobs;
};
struct Pending {
  float X[3];
  Obs o;
  uint32_t ok;  // the ADD solve of this candidate succeeded
  uint32_t pad;
};
// Per chain point, for the view being offered: unique 4 px-grid polyline and closest point.
struct ViewCand {
  uint32_t valid;  // exactly one polyline in the window
  uint32_t pl, seg;
  float x, y, d2;
};

// Execution team of one chain. The expand stage is written SPMD-style: every member runs the
// same control flow on identical values ("uniform" sections, redundant across lanes) and only
// the marked parallel sections split items across members, communicating through the chain's
// scratch slice followed by sync(). TeamSeq (1 member) is the sequential semantics and what the
// host instantiation uses; the GPU kernel uses a 64-lane wavefront (TeamWave in
// eg3d_kernels.hip).
struct TeamSeq {
  EG3D_HD int lane() const { return 0; }
  EG3D_HD int size() const { return 1; }
  EG3D_HD void sync() const {}
};

// Per-chain working set (pointers into this chain's scratch slice).
struct Chain {
  ChainPt* pts;
  int32_t cap_pts;
  int32_t head;  // index of chain[0] in pts
  int32_t len;
  PoolObs* pool;
  uint32_t pool_used, pool_cap;
  uint32_t* start_dirs;  // [V] node id the chain front is heading to, per view
  uint32_t* end_dirs;    // [V]
  Pending* pend1;        // [cap_pts]
  Pending* pend2;        // [cap_pts]
  ViewCand* cand;        // [cap_pts]
  Obs* tmp_a;            // [tmp_cap]
  Obs* tmp_b;            // [tmp_cap]
  uint8_t* tmp_mask;     // [tmp_cap]
  int32_t tmp_cap;
  uint32_t flags;
  uint64_t bytes;  // vertices of 4 px-grid polylines tested (algorithmic bytes)
};

EG3D_HD ChainPt& chain_at(Chain& c, int i) { return c.pts[c.head + i]; }

struct ListCursor {
  const PoolObs* pool;
  uint32_t head;
  int n;
  const Obs* extra;
  uint32_t cur;
  int i;
  EG3D_HD void rewind() {
    cur = head;
    i = 0;
  }
  EG3D_HD int count() const { return n + (extra ? 1 : 0); }
  EG3D_HD bool next(int32_t& view, float& x, float& y) {
    if (i < n) {
      const PoolObs& p = pool[cur];
      view = p.o.view;
      x = p.o.x;
      y = p.o.y;
      cur = p.next;
      i++;
      return true;
    }
    if (extra && i == n) {
      view = extra->view;
      x = extra->x;
      y = extra->y;
      i++;
      return true;
    }
    return false;
  }
};

EG3D_HD bool pool_append(Chain& c, ChainPt& p, const Obs& o) {
  if (c.pool_used >= c.pool_cap) {
    c.flags |= 2u;
    return false;
  }
  uint32_t idx = c.pool_used++;
  c.pool[idx].o = o;
  c.pool[idx].next = 0xffffffffu;
  if (p.nobs == 0)
    p.head = idx;
  else
    c.pool[p.tail].next = idx;
  p.tail = idx;
  p.nobs++;
  return true;
}

// GN from the stored X with the point's observations plus one more (ADD).
EG3D_HD bool add_observation_solve(const DevScene& s, const Chain& c, const ChainPt& p, const Obs& extra,
                                   float Xout[3]) {
  ListCursor cur;
  cur.pool = c.pool;
  cur.head = p.head;
  cur.n = (int)p.nobs;
  cur.extra = &extra;
  cur.rewind();
  double X0[3] = {(double)p.X[0], (double)p.X[1], (double)p.X[2]};
  return gauss_newton_f64(s.cam_P, cur, X0, Xout);
}

// N-view step on a chain point with any number of observations; the new point is returned in
// tmp_a[0..m) with its X. Returns the number of observations of the new point (0 = failure).
EG3D_HD int stepn_chain(const DevScene& s, Chain& c, const ChainPt& cur, const uint32_t* dirs, float Xout[3]) {
  const int n = (int)cur.nobs;
  uint32_t si = cur.head;
  for (int st = 0; st < n; st++, si = c.pool[si].next) {
    const Obs so = c.pool[si].o;
    PlRef ps = polyline_of(s, so.view, so.pl);
    PlPt p, q;
    p.seg = so.seg;
    p.x = so.x;
    p.y = so.y;
    uint32_t w = walk_by_distance(ps, p, dirs[so.view], EG3D_FOLLOW_STEP, q);
    if (w & WALK_BAD_DIR) c.flags |= 8u;
    if (w & WALK_EXTREME) continue;
    Obs* sel = c.tmp_a;
    int m = 0;
    sel[m].view = so.view;
    sel[m].pl = so.pl;
    sel[m].seg = q.seg;
    sel[m].x = q.x;
    sel[m].y = q.y;
    m++;
    uint32_t oi = cur.head;
    for (int i = 0; i < n; i++, oi = c.pool[oi].next) {
      if (i == st) continue;
      const Obs co = c.pool[oi].o;
      float la, lb, lc;
      if (!epiline(s.F, s.F_valid, s.n_views, so.view, co.view, q.x, q.y, la, lb, lc)) continue;
      PlRef pk = polyline_of(s, co.view, co.pl);
      PlPt cp, r;
      cp.seg = co.seg;
      cp.x = co.x;
      cp.y = co.y;
      uint32_t wr = walk_by_line(pk, cp, dirs[co.view], la, lb, lc, true, EG3D_FOLLOW_MIN, EG3D_FOLLOW_MAX, r);
      if (wr & WALK_BAD_DIR) c.flags |= 8u;
      if (wr & WALK_FOUND) {
        if (m >= c.tmp_cap) {
          c.flags |= 2u;
          break;
        }
        sel[m].view = co.view;
        sel[m].pl = co.pl;
        sel[m].seg = r.seg;
        sel[m].x = r.x;
        sel[m].y = r.y;
        m++;
      }
    }
    if (m < 3) continue;
    bool valid = triangulate_array(s.cam_P, sel, m, Xout, c.flags);
    if (!valid && m > 3) {
      valid = triangulate_combinations(s.cam_P, sel, m, c.tmp_b, c.tmp_mask, Xout, c.flags);
      if (valid) {
        int k = 0;
        for (int i = 0; i < m; i++)
          if (c.tmp_mask[i]) sel[k++] = sel[i];
        m = k;
      }
    }
    if (valid) return m;
  }
  return 0;
}

EG3D_HD bool new_point_from_tmp(Chain& c, ChainPt& np, int m, const float X[3]) {
  np.X[0] = X[0];
  np.X[1] = X[1];
  np.X[2] = X[2];
  np.nobs = 0;
  np.head = 0xffffffffu;
  np.tail = 0xffffffffu;
  for (int i = 0; i < m; i++)
    if (!pool_append(c, np, c.tmp_a[i])) return false;
  return true;
}

// Grow the chain at the back / at the front while steps succeed. Returns points added.
EG3D_HD int follow_back(const DevScene& s, Chain& c) {
  int added = 0;
  for (;;) {
    float X[3];
    int m = stepn_chain(s, c, chain_at(c, c.len - 1), c.end_dirs, X);
    if (m == 0) break;
    if (c.head + c.len >= c.cap_pts) {
      c.flags |= 1u;
      break;
    }
    ChainPt np;
    if (!new_point_from_tmp(c, np, m, X)) break;
    c.pts[c.head + c.len] = np;
    c.len++;
    added++;
  }
  return added;
}
EG3D_HD int follow_front(const DevScene& s, Chain& c) {
  int added = 0;
  for (;;) {
    float X[3];
    int m = stepn_chain(s, c, chain_at(c, 0), c.start_dirs, X);
    if (m == 0) break;
    if (c.head <= 0) {
      c.flags |= 1u;
      break;
    }
    ChainPt np;
    if (!new_point_from_tmp(c, np, m, X)) break;
    c.head--;
    c.pts[c.head] = np;
    c.len++;
    added++;
  }
  return added;
}

// Side walk, phase 1 (uniform): from `from` on view `view` towards node `direction`, visit chain
// points ci-1, ci-2, ... >= lo (towards_start) or ci+1, ... < hi; for each take the next hit of
// the epipolar line of the point's FIRST observation. Walk positions do not depend on the
// solver, so all candidates are generated first; returns how many walks succeeded.
EG3D_HD int walk_side_candidates(const DevScene& s, Chain& c, int view, const Obs& from, uint32_t direction, int lo,
                                 int ci, int hi, bool towards_start, Pending* out) {
  int cnt = 0;
  PlRef pl = polyline_of(s, view, from.pl);
  PlPt actual;
  actual.seg = from.seg;
  actual.x = from.x;
  actual.y = from.y;
  int i = towards_start ? ci - 1 : ci + 1;
  while ((towards_start && i >= lo) || (!towards_start && i < hi)) {
    const ChainPt& pt = chain_at(c, i);
    const Obs& first = c.pool[pt.head].o;
    float la, lb, lc;
    if (!epiline(s.F, s.F_valid, s.n_views, first.view, view, first.x, first.y, la, lb, lc)) break;
    PlPt nx;
    uint32_t w = walk_by_line(pl, actual, direction, la, lb, lc, false, 0.0f, 0.0f, nx);
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

// Side walk, phase 2 (PARALLEL over candidates): ADD-solve candidate j against chain point
// ci-1-j / ci+1+j; then (uniform) count the leading successes — the reference stops at the
// first failed walk or solve (plg_matching.cpp:866-914).
template <class Team>
EG3D_HD int walk_side(const Team& tm, const DevScene& s, Chain& c, int view, const Obs& from, uint32_t direction,
                      int lo, int ci, int hi, bool towards_start, Pending* out) {
  const int m = walk_side_candidates(s, c, view, from, direction, lo, ci, hi, towards_start, out);
  tm.sync();
  for (int j = tm.lane(); j < m; j += tm.size()) {
    const ChainPt& pt = chain_at(c, towards_start ? ci - 1 - j : ci + 1 + j);
    float X[3];
    const Obs o = out[j].o;
    if (add_observation_solve(s, c, pt, o, X)) {
      out[j].X[0] = X[0];
      out[j].X[1] = X[1];
      out[j].X[2] = X[2];
      out[j].ok = 1;
    }
  }
  tm.sync();
  int cnt = 0;
  while (cnt < m && out[cnt].ok) cnt++;
  return cnt;
}

// Try to attach observation `o` of view o.view to chain point ci, then to its neighbours within
// [lo, hi). On success returns true with (to_start, to_end) = observations added on each side
// including newly grown points. (add_view_to_3dpoint_and_sides_plgp_matches_vector, Q13.)
template <class Team>
EG3D_HD bool attach_view(const Team& tm, const DevScene& s, Chain& c, const Obs& o, int lo, int ci, int hi,
                         int& to_start, int& to_end) {
  to_start = 0;
  to_end = 0;
  float Xc[3];
  if (!add_observation_solve(s, c, chain_at(c, ci), o, Xc)) return false;
  const int view = o.view;
  PlRef pl = polyline_of(s, view, o.pl);
  uint32_t nd1 = 0, nd2 = 0;
  int n1 = 0, n2 = 0;
  if (ci > lo) {
    n1 = walk_side(tm, s, c, view, o, pl.start, lo, ci, hi, true, c.pend1);
    if (n1 > 0) {
      nd1 = pl.start;
      nd2 = pl.end;
      if (ci < hi) n2 = walk_side(tm, s, c, view, o, pl.end, lo, ci, hi, false, c.pend2);
    } else {
      n1 = walk_side(tm, s, c, view, o, pl.end, lo, ci, hi, true, c.pend1);
      if (n1 > 0) {
        nd1 = pl.end;
        nd2 = pl.start;
        if (ci < hi) n2 = walk_side(tm, s, c, view, o, pl.start, lo, ci, hi, false, c.pend2);
      }
      // else: neither orientation reaches the lower neighbour; with ci > lo >= 0 the
      // attachment is rejected below whatever the upper side would give.
    }
  }
  if (ci > 0 && n1 == 0) return false;
  if (ci < c.len - 1 && n2 == 0) return false;
  // commit (uniform: every member writes the same values)
  {
    ChainPt& cp = chain_at(c, ci);
    cp.X[0] = Xc[0];
    cp.X[1] = Xc[1];
    cp.X[2] = Xc[2];
    pool_append(c, cp, o);
  }
  for (int i = 0; i < n1; i++) {
    ChainPt& p = chain_at(c, ci - 1 - i);
    p.X[0] = c.pend1[i].X[0];
    p.X[1] = c.pend1[i].X[1];
    p.X[2] = c.pend1[i].X[2];
    pool_append(c, p, c.pend1[i].o);
  }
  for (int i = 0; i < n2; i++) {
    ChainPt& p = chain_at(c, ci + 1 + i);
    p.X[0] = c.pend2[i].X[0];
    p.X[1] = c.pend2[i].X[1];
    p.X[2] = c.pend2[i].X[2];
    pool_append(c, p, c.pend2[i].o);
  }
  to_start = n1;
  to_end = n2;
  if (n1 > 0 && n1 == ci) {
    c.start_dirs[view] = nd1;
    int g = follow_front(s, c);
    to_start += g;
    ci += g;
  }
  if (n2 > 0 && n2 == (c.len - ci - 1)) {
    c.end_dirs[view] = nd2;
    to_end += follow_back(s, c);
  }
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

// PARALLEL over chain points [from, len): project the point into view v, look up the unique
// polyline of the 4 px grid and its closest point (triangulation.cpp:791-806).
template <class Team>
EG3D_HD void view_candidates(const Team& tm, const DevScene& s, Chain& c, int v, int from) {
  const float* P = s.cam_P + (size_t)v * 16;
  for (int i = from + tm.lane(); i < c.len; i += tm.size()) {
    const ChainPt& pt = chain_at(c, i);
    ViewCand vc;
    vc.valid = 0;
    vc.pl = 0;
    vc.seg = 0;
    vc.x = vc.y = vc.d2 = 0.0f;
    float u, w;
    project_f32(P, pt.X[0], pt.X[1], pt.X[2], u, w);
    uint32_t pl_id;
    if (unique_polyline_4px(s, v, u, w, pl_id)) {
      PlRef pl = polyline_of(s, v, pl_id);
      PlPt cp;
      vc.d2 = polyline_closest(pl, u, w, cp);
      vc.valid = 1;
      vc.pl = pl_id;
      vc.seg = cp.seg;
      vc.x = cp.x;
      vc.y = cp.y;
    }
    c.cand[i] = vc;
  }
  tm.sync();
}

// Offer the chain to view v. epc = the task's epipolar hits in v (may be empty).
template <class Team>
EG3D_HD void expand_to_view(const Team& tm, const DevScene& s, Chain& c, int v, const Obs* epc, int n_epc,
                            int& centre) {
  bool epc_matched = false;
  int idx_first = 0, idx_second = 0;
  for (int e = 0; e < n_epc; e++) {
    int a, b;
    if (attach_view(tm, s, c, epc[e], 0, centre, c.len, a, b)) {
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
  for (int cur = 0; cur < c.len; cur++) {
    if (epc_matched && cur == idx_first) {
      cur = idx_second;
      last_matched = idx_second;
      continue;
    }
    const ViewCand vc = c.cand[cur];
    if (!vc.valid) continue;
    c.bytes += 8ull * (s.pl_vtx_off[s.view_pl_off[v] + vc.pl + 1] - s.pl_vtx_off[s.view_pl_off[v] + vc.pl]);
    if (vc.d2 > 16.0f) return;  // abandons this view (Q4)
    Obs o;
    o.view = v;
    o.pl = vc.pl;
    o.seg = vc.seg;
    o.x = vc.x;
    o.y = vc.y;
    int hi = epc_matched ? (cur <= idx_first ? idx_first : c.len) : c.len;
    int a, b;
    if (attach_view(tm, s, c, o, last_matched + 1, cur, hi, a, b)) {
      if (a > cur) {
        centre = a;
        cur = a + b;
      } else {
        cur = cur + b;
      }
      last_matched = cur;
      // the chain may have grown / shifted: refresh the candidates of the points still to visit
      view_candidates(tm, s, c, v, cur + 1);
    }
  }
}

}  // namespace eg3d
