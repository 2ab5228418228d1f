// eg3d_chain_sm.h — the expand-all-views stage as a per-chain STATE MACHINE (host + device).
//
// eg3d_dev_expand.h states the stage as nested calls executed by a team that owns ONE chain (the round 1-4 kernel: one
// wavefront per chain, k3b_expand). This file states the SAME program — same walks, same solves, same order, same
// flags, same bytes — as a resumable machine whose only blocking points are two kinds of batch request:
//
//   GN       a batch of independent Gauss-Newton solves (ADD solves of a point's block plus one observation, or a solve
//            over an observation list), answered through memory of the chain's working slice;
//   CLOSEST  the per-view candidate of a range of chain points (projection, 4 px-grid lookup, closest point, epipolar
//            line of the point's first observation).
//
// Everything between two requests is lane-private work (walks, the 2-view DLT, commits, bookkeeping). That is what lets
// ONE LANE own a chain: a wavefront of the engine kernel (eg3d_k3c_engine.h) advances 64 machines, then serves all
// their requests densely — rows of all chains' solves side by side over the 64 lanes, candidate items dealt to lanes —
// instead of 64 lanes executing one chain's wave-uniform sections redundantly. The host simulation of the tests drives
// the same machine with a sequential server (tests/hostsim, mode 2), which is how its logic is pinned against the
// oracle on the CPU.
//
// Reference behaviour (as eg3d_dev_expand.h): expand_allpoints_to_other_view_using_plmap (triangulation.cpp:742-833),
// add_view_to_3dpoint_and_sides_plgp_matches_vector, compatible_direction_noupdate_vector, follow_direction_vector_start /
// _end, compatible() vector form (plg_matching.cpp:1345-1412, 866-914, 771-795, 633-759), the 3-subset fallback
// (triangulation.cpp:1105-1158).
#pragma once
#include "eg3d_dev_pipeline.h"

namespace eg3d {

// result of one solve, in the chain's slice (single solves of the machine: central, N-view step, fallback)
struct SmMbox {
  uint32_t ok;
  float X[3];
};

enum : uint32_t { SM_RUN = 0, SM_WAIT_GN = 1, SM_WAIT_CL = 2, SM_DONE = 3 };
enum : uint32_t {
  SMB_EPC = 1,       // request e: chain[centre] + epc[e]            -> epcres[e]
  SMB_PRESOLVE = 2,  // request j: chain[from + j] + its candidate   -> cand[head + from + j].cok / cX (skipped: no candidate within 4 px)
  SMB_CENTRAL = 3,   // chain[ci] + o                                 -> mbox
  SMB_SIDES = 4,     // j < m1: chain[ci-1-j] + pend1[j].o, else chain[ci+1+(j-m1)] + pend2[j-m1].o -> pendX[j].X / .ok
  SMB_LIST_A = 5,    // rows tmp_a[0..m) from X (the N-view step's candidate list) -> mbox
  SMB_LISTADD = 6,   // rows tmp_b[0..kept) + tmp_a[fb_i] from X (greedy phase of the fallback) -> mbox
  SMB_LIST_B = 7     // rows tmp_b[0..3) from X (a 3-subset of the fallback)     -> mbox
};
enum : uint32_t {
  SMS_VIEW_NEXT = 0,
  SMS_EPC_POST,
  SMS_EPC_LOOP,
  SMS_CAND_POST,
  SMS_CAND_DONE,
  SMS_VISIT,
  SMS_VISIT_ATTACH,
  SMS_ATTACH_BEGIN,
  SMS_CENTRAL_DONE,
  SMS_ATTACH_SIDES,
  SMS_SIDES_DONE,
  SMS_ATTACH_CHECK,
  SMS_FOLLOW_SIDE,
  SMS_FOLLOW_STEP,
  SMS_STEP_CAND,
  SMS_STEP_TRI,
  SMS_STEP_TRI_DONE,
  SMS_FB_NEXT,
  SMS_FB_TRI_DONE,
  SMS_FB_ADD,
  SMS_FB_ADD_DONE,
  SMS_STEP_OK,
  SMS_FOLLOW_END,
  SMS_ATTACH_RET,
  SMS_FINISH
};
enum : uint32_t { SMR_EPC = 0, SMR_VISIT = 1 };

// What a chain is made from (constant for its life).
struct SmTask {
  int32_t sel_view[3];  // the three views of the hypothesis (not offered)
  const int32_t* mv;    // the seed's view map
  const uint32_t* me;
  uint32_t n_map, list_lo;
};

// Control state of one machine.
struct SmCtl {
  uint32_t st, wait;
  // pending GN batch
  uint32_t gn_kind;
  int32_t gn_count, gn_issued;
  int32_t gn_from;        // PRESOLVE: first chain point of the window
  // pending CLOSEST batch: chain points [cl_from, cl_to) in view v; cl_epi_only: just the epipolar lines of the points'
  // first observations (before the epipolar hits of a view are tried), not the candidates
  int32_t cl_from, cl_to;
  uint32_t cl_epi_only;
  // view loop
  int32_t v;
  uint32_t j;
  const Obs* epc;
  int32_t n_epc, n_pre, e;
  int32_t centre;
  // expand_to_view
  uint32_t epc_matched;
  int32_t idx_first, idx_second, last_matched, cur, spec_slot_hi;
  // attach_view
  Obs o;
  int32_t lo, ci, hi;
  uint32_t ret_to, pre_kind;  // pre_kind: 0 = no speculative central solve, 1 = epcres[e], 2 = cand[head + cur]
  int32_t which, n1, n2, m1, m2;
  uint32_t nd1, nd2;
  int32_t to_start, to_end;
  uint32_t attach_ok;
  // following
  int32_t side, added, st_obs, m;
  // attach_view: the central solve's point (until the commit); following: start point / result of the step's solve and
  // the fallback's running point (after the commit) — never live together
  float X[3];
  // 3-subset fallback
  int32_t fi, fj, fk, kept, fb_i;
};

struct SmChain {
  Chain c;
  SmCtl k;
  SmTask t;
  SmMbox* mbox;
};

EG3D_HD void sm_post_gn(SmCtl& k, uint32_t kind, int count, uint32_t next_state) {
  k.gn_kind = kind;
  k.gn_count = count;
  k.gn_issued = 0;
  k.st = next_state;
  if (count > 0) k.wait = SM_WAIT_GN;
}

// One request of the pending GN batch, as the server sees it. Returns false for a request the batch skips.
struct SmGnReq {
  const Obs* base;
  int32_t nblock;
  uint32_t has_extra;
  int32_t ex_view;
  float ex_x, ex_y;
  float X0[3];
  float* resX;       // where the solution goes (3 floats) ...
  uint32_t* resOk;   // ... and the verdict
};
EG3D_HD bool sm_gn_request(SmChain& q, int j, SmGnReq& r) {
  Chain& c = q.c;
  const SmCtl& k = q.k;
  const ChainPt* pt = nullptr;
  r.has_extra = 1;
  switch (k.gn_kind) {
    case SMB_EPC: {
      pt = &chain_at(c, k.centre);
      const Obs o = k.epc[j];
      r.ex_view = (int32_t)o.view;
      r.ex_x = o.x;
      r.ex_y = o.y;
      r.resX = c.epcres[j].X;
      r.resOk = &c.epcres[j].ok;
      break;
    }
    case SMB_PRESOLVE: {
      ViewCand& vc = c.cand[c.head + k.gn_from + j];
      if (!vc.valid || !(vc.d2 <= 16.0f)) return false;
      pt = &chain_at(c, k.gn_from + j);
      r.ex_view = k.v;
      r.ex_x = vc.x;
      r.ex_y = vc.y;
      r.resX = vc.cX;
      r.resOk = &vc.cok;
      break;
    }
    case SMB_CENTRAL: {
      pt = &chain_at(c, k.ci);
      r.ex_view = (int32_t)k.o.view;
      r.ex_x = k.o.x;
      r.ex_y = k.o.y;
      r.resX = q.mbox->X;
      r.resOk = &q.mbox->ok;
      break;
    }
    case SMB_SIDES: {
      Pending* pd;
      if (j < k.m1) {
        pt = &chain_at(c, k.ci - 1 - j);
        pd = &c.pend1[j];
      } else {
        pt = &chain_at(c, k.ci + 1 + (j - k.m1));
        pd = &c.pend2[j - k.m1];
      }
      r.ex_view = (int32_t)pd->o.view;
      r.ex_x = pd->o.x;
      r.ex_y = pd->o.y;
      r.resX = pd->X;
      r.resOk = &pd->ok;
      break;
    }
    case SMB_LIST_A:
    case SMB_LIST_B:
    case SMB_LISTADD: {
      r.base = k.gn_kind == SMB_LIST_A ? c.tmp_a : c.tmp_b;
      r.nblock = k.gn_kind == SMB_LIST_A ? k.m : k.gn_kind == SMB_LIST_B ? 3 : k.kept;
      r.has_extra = k.gn_kind == SMB_LISTADD ? 1u : 0u;
      r.ex_view = 0;
      r.ex_x = r.ex_y = 0.0f;
      if (k.gn_kind == SMB_LISTADD) {
        const Obs ex = c.tmp_a[k.fb_i];
        r.ex_view = (int32_t)ex.view;
        r.ex_x = ex.x;
        r.ex_y = ex.y;
      }
      r.X0[0] = k.X[0];
      r.X0[1] = k.X[1];
      r.X0[2] = k.X[2];
      r.resX = q.mbox->X;
      r.resOk = &q.mbox->ok;
      return true;
    }
    default:
      return false;
  }
  r.base = c.pool + pt->off;
  r.nblock = (int32_t)pt->nobs;
  r.X0[0] = pt->X[0];
  r.X0[1] = pt->X[1];
  r.X0[2] = pt->X[2];
  return true;
}

// CLOSEST item: the candidate of chain point i in view v (view_candidates of eg3d_dev_expand.h with one member per
// point), stored in the candidate array.
EG3D_HD void sm_closest_item(const DevScene& s, const ChainPt* pts, const Obs* pool, ViewCand* cand, int head, int v, int i) {
  const float* P = s.cam_P + (size_t)v * 16;
  ViewCand vc;
  vc.valid = 0;
  vc.pl = 0;
  vc.seg = 0;
  vc.x = vc.y = vc.d2 = 0.0f;
  vc.cok = 0;
  vc.cX[0] = vc.cX[1] = vc.cX[2] = 0.0f;
  vc.eok = 0;
  vc.ea = vc.eb = vc.ec = 0.0f;
  const ChainPt pt = pts[head + i];
  const Obs first = pool[pt.off];
  vc.eok = epiline(s.F, s.F_valid, s.n_views, first.view, v, first.x, first.y, vc.ea, vc.eb, vc.ec) ? 1u : 0u;
  float u, w;
  project_f32(P, pt.X[0], pt.X[1], pt.X[2], u, w);
  uint32_t pl_id;
  if (unique_polyline_4px(s, v, u, w, pl_id)) {
    PlRef pl = polyline_of_bb(s, v, pl_id);
    PlPt cp;
    cp.seg = 0xffffffffu;
    cp.x = cp.y = 0.0f;
    const float d2 = polyline_closest_pruned(pl, u, w, 0u, pl.n - 1u, cp);
    vc.valid = 1;
    vc.pl = pl_id;
    vc.d2 = d2;
    vc.seg = cp.seg;
    vc.x = cp.x;
    vc.y = cp.y;
  }
  cand[head + i] = vc;
}
EG3D_HD void sm_closest_item(const DevScene& s, Chain& c, int v, int i) { sm_closest_item(s, c.pts, c.pool, c.cand, c.head, v, i); }
// ... only the epipolar line of the point's first observation (the other fields of the candidate are left alone)
EG3D_HD void sm_epiline_item(const DevScene& s, const ChainPt* pts, const Obs* pool, ViewCand* cand, int head, int v, int i) {
  const Obs first = pool[pts[head + i].off];
  float ea = 0.0f, eb = 0.0f, ec = 0.0f;
  const uint32_t eok = epiline(s.F, s.F_valid, s.n_views, first.view, v, first.x, first.y, ea, eb, ec) ? 1u : 0u;
  ViewCand& vc = cand[head + i];
  vc.eok = eok;
  vc.ea = ea;
  vc.eb = eb;
  vc.ec = ec;
}

// Build the initial chain (expand_chain's first part) and put the machine before its first view.
EG3D_HD void sm_begin(const DevScene& s, const StageAView& a, const TaskDesc& d, const ChainSeed& cs, uint32_t hyp_base,
                      const HypResult* res, const HPoint* arena, const int32_t* map_view, const uint32_t* map_entry,
                      const uint32_t* map_n, const ChainLayout& L, unsigned char* slice, SmMbox* mbox, SmChain& q) {
  Chain& c = q.c;
  chain_bind(c, L, slice);
  q.mbox = mbox;
  c.flags = 0;
  c.bytes = 0;
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
    for (int i0 = 0; i0 < L1; i0 += 4) {  // four points requested together (the lane builds its chain alone)
      HPoint hp[4];
      for (int b = 0; b < 4; b++) {
        const int i = i0 + b < L1 ? i0 + b : i0;
        if (i < centre0) {
          hp[b] = arena[w.pts1_off + (uint32_t)(centre0 - 1 - i)];
        } else if (i == centre0) {
          hypothesis_hits(a, d, cs.task, cs.winner - hyp_base, hp[b].o);
          hp[b].X[0] = w.X[0];
          hp[b].X[1] = w.X[1];
          hp[b].X[2] = w.X[2];
          hp[b].nobs = 3;
          hp[b].pad = 0;
        } else {
          hp[b] = arena[p2 + (uint32_t)(i - centre0 - 1)];
        }
      }
      for (int b = 0; b < 4 && i0 + b < L1; b++) {
        const int i = i0 + b;
        ChainPt p;
        p.X[0] = hp[b].X[0];
        p.X[1] = hp[b].X[1];
        p.X[2] = hp[b].X[2];
        p.off = 4u * (uint32_t)i;
        p.cap = 4;
        p.nobs = hp[b].nobs;
        for (uint32_t k = 0; k < hp[b].nobs; k++) c.pool[p.off + k] = hp[b].o[k];
        c.pts[c.head + i] = p;
      }
    }
    c.len = L1;
    c.pool_used = 4u * (uint32_t)L1;
  }
  SmTask& t = q.t;
  t.sel_view[0] = d.sel_view[0];
  t.sel_view[1] = d.sel_view[1];
  t.sel_view[2] = d.sel_view[2];
  const uint32_t base = track_base(a, d.seed);
  t.n_map = track_n_views(a, map_n, d.seed);
  t.mv = map_view + base;
  t.me = map_entry + base;
  t.list_lo = a.task_list_off[cs.task];
  SmCtl& k = q.k;
  k.st = SMS_VIEW_NEXT;
  k.wait = SM_RUN;
  k.gn_kind = 0;
  k.gn_count = k.gn_issued = 0;
  k.gn_from = 0;
  k.o.view = 0;
  k.o.pl = 0;
  k.o.seg = 0;
  k.o.x = k.o.y = 0.0f;
  k.X[0] = k.X[1] = k.X[2] = 0.0f;
  k.cl_from = k.cl_to = 0;
  k.cl_epi_only = 0;
  k.v = -1;
  k.j = 0;
  k.epc = nullptr;
  k.n_epc = k.n_pre = k.e = 0;
  k.centre = centre0;
  k.epc_matched = 0;
  k.idx_first = k.idx_second = 0;
  k.last_matched = -1;
  k.cur = 0;
  k.spec_slot_hi = 0;
  k.lo = k.ci = k.hi = 0;
  k.ret_to = 0;
  k.pre_kind = 0;
  k.which = k.n1 = k.n2 = k.m1 = k.m2 = 0;
  k.nd1 = k.nd2 = 0;
  k.to_start = k.to_end = 0;
  k.attach_ok = 0;
  k.side = k.added = k.st_obs = k.m = 0;
  k.fi = k.fj = k.fk = k.kept = k.fb_i = 0;
}

// dst[0..n) = src[0..n) (disjoint), four observations requested at a time
EG3D_HD void sm_copy_obs(Obs* dst, const Obs* src, uint32_t n) {
  uint32_t i = 0;
  for (; i + 4 <= n; i += 4) {
    const Obs a0 = src[i], a1 = src[i + 1], a2 = src[i + 2], a3 = src[i + 3];
    dst[i] = a0;
    dst[i + 1] = a1;
    dst[i + 2] = a2;
    dst[i + 3] = a3;
  }
  for (; i < n; i++) dst[i] = src[i];
}
// leading candidates of a side whose solve succeeded; the verdicts are requested eight at a time (a lane scans alone:
// one trip to memory per eight candidates instead of one each)
EG3D_HD int sm_leading_ok(const Pending* pd, int m) {
  int cnt = 0;
  while (cnt < m) {
    uint32_t ok[8];
    for (int b = 0; b < 8; b++) ok[b] = cnt + b < m ? pd[cnt + b].ok : 0u;
    int b = 0;
    while (b < 8 && ok[b] != 0) b++;
    cnt += b;
    if (b < 8) break;
  }
  return cnt;
}

// presolve policy of the machine: windows of points as the visit reaches them (lazy) or the whole view at once
EG3D_HD bool sm_lazy_presolve(const DevScene& s) { return lazy_presolve(s); }

// min-view / last entry of a list: the two observations of its initial DLT (triangulate_array); raises flag 16 when
// they are of the same view
template <class Env>
EG3D_HD void sm_list_dlt(const Env& env, const DevScene& s, const Obs* a, int n, uint32_t& flags, float X0f[3]) {
  int mi = 0;
  int32_t mv = (int32_t)a[0].view;
  for (int i = 0; i < n; i++)
    if ((int32_t)a[i].view < mv) {
      mv = (int32_t)a[i].view;
      mi = i;
    }
  const int la = n - 1;
  if (a[mi].view == a[la].view) flags |= 16u;
  const Obs o1 = a[mi], o2 = a[la];
  double X0[3];
  env.dlt(s.cam_P + (size_t)o1.view * 16, o1.x, o1.y, s.cam_P + (size_t)o2.view * 16, o2.x, o2.y, X0);
  X0f[0] = (float)X0[0];  // (DLT results are float-valued)
  X0f[1] = (float)X0[1];
  X0f[2] = (float)X0[2];
}

// The side walks of one orientation (walk_sides_both's `walks`).
template <class Env>
EG3D_HD void sm_walks(const Env& env, const DevScene& s, SmChain& q, const PlRef& pl, uint32_t dS, uint32_t dE) {
  Chain& c = q.c;
  SmCtl& k = q.k;
  k.m1 = env.side_walk(s, c, (int)k.o.view, pl, k.o, dS, k.lo, k.ci, k.hi, true, c.pend1);
  k.m2 = 0;
  if (k.m1 > 0 && k.ci < k.hi) k.m2 = env.side_walk(s, c, (int)k.o.view, pl, k.o, dE, k.lo, k.ci, k.hi, false, c.pend2);
}

// timing builds of the engine kernel: clocks per block of sm_advance (the Env counts; the plain Envs do nothing)
template <class Env>
struct SmProf {
  const Env& e;
  uint32_t id;
  EG3D_HD SmProf(const Env& env, uint32_t i) : e(env), id(i) { e.prof_begin(id); }
  EG3D_HD ~SmProf() { e.prof_end(id); }
};
// Run the machine until it blocks (k.wait != SM_RUN). The blocks below are in flow order, so that a machine passes
// through as many of them as it can in one trip of the loop (on the GPU a trip executes every block some lane is in).
template <class Env>
EG3D_HD_FLAT void sm_advance(const Env& env, const DevScene& s, const StageAView& a, SmChain& q) {
  Chain& c = q.c;
  SmCtl& k = q.k;
  const SmTask& t = q.t;
  while (k.wait == SM_RUN) {
    // ---------------- result of an attachment, back in the loop that tried it
    if (k.st == SMS_ATTACH_RET) {
      const SmProf<Env> prof_(env, SMS_ATTACH_RET);
      if (k.ret_to == SMR_EPC) {
        if (k.attach_ok) {
          k.epc_matched = 1;
          const int aa = k.to_start, bb = k.to_end;
          if (aa > k.centre) {
            k.centre = aa;
            k.idx_first = 0;
            k.idx_second = aa + bb;
          } else {
            k.idx_first = k.centre - aa;
            k.idx_second = k.centre + bb;
          }
          k.st = SMS_CAND_POST;
        } else {
          k.e++;
          k.st = SMS_EPC_LOOP;
        }
      } else {
        if (k.attach_ok) {
          const int aa = k.to_start, bb = k.to_end;
          if (aa > k.cur) {
            k.centre = aa;
            k.cur = aa + bb;
          } else {
            k.cur = k.cur + bb;
          }
          k.last_matched = k.cur;
        }
        k.cur++;
        k.st = SMS_VISIT;
      }
    }
    // ---------------- next view
    if (k.st == SMS_VIEW_NEXT) {
      const SmProf<Env> prof_(env, SMS_VIEW_NEXT);
      for (;;) {
        k.v++;
        if (k.v >= s.n_views) break;
        if (k.v == t.sel_view[0] || k.v == t.sel_view[1] || k.v == t.sel_view[2]) continue;
        break;
      }
      if (k.v >= s.n_views) {
        k.st = SMS_FINISH;
      } else {
        while (k.j < t.n_map && t.mv[k.j] < k.v) k.j++;
        k.epc = nullptr;
        k.n_epc = 0;
        if (k.j < t.n_map && t.mv[k.j] == k.v) {
          k.epc = a.hits + a.list_ptr[t.list_lo + t.me[k.j]];
          k.n_epc = (int)a.list_cnt[t.list_lo + t.me[k.j]];
        }
        k.epc_matched = 0;
        k.idx_first = k.idx_second = 0;
        k.n_pre = k.n_epc < c.cap_pts ? k.n_epc : c.cap_pts;
        k.e = 0;
        if (k.n_epc > 0) {
          // epipolar lines of every chain point in view v (the side walks read them): an epi-only CLOSEST batch, then
          // the speculative central solves of the view's epipolar hits
          k.cl_from = 0;
          k.cl_to = c.len;
          k.cl_epi_only = 1;
          k.st = SMS_EPC_POST;
          if (k.cl_to > k.cl_from) k.wait = SM_WAIT_CL;
        } else {
          k.st = SMS_CAND_POST;
        }
      }
    }
    if (k.st == SMS_EPC_POST && k.wait == SM_RUN) sm_post_gn(k, SMB_EPC, k.n_pre, SMS_EPC_LOOP);
    // ---------------- the task's epipolar hits in this view, in order, against the central point
    if (k.st == SMS_EPC_LOOP && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_EPC_LOOP);
      if (k.e >= k.n_epc) {
        k.st = SMS_CAND_POST;
      } else {
        k.o = k.epc[k.e];
        k.lo = 0;
        k.ci = k.centre;
        k.hi = c.len;
        k.pre_kind = k.e < k.n_pre ? 1u : 0u;
        k.ret_to = SMR_EPC;
        k.st = SMS_ATTACH_BEGIN;
      }
    }
    // ---------------- candidates of all chain points in view v
    if (k.st == SMS_CAND_POST) {
      const SmProf<Env> prof_(env, SMS_CAND_POST);
      k.last_matched = -1;
      k.cl_from = 0;
      k.cl_to = c.len;
      k.cl_epi_only = 0;
      k.st = SMS_CAND_DONE;
      if (k.cl_to > k.cl_from) k.wait = SM_WAIT_CL;
    }
    if (k.st == SMS_CAND_DONE && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_CAND_DONE);
      k.spec_slot_hi = 0;
      k.cur = 0;
      if (!sm_lazy_presolve(s)) {
        k.gn_from = 0;
        sm_post_gn(k, SMB_PRESOLVE, c.len, SMS_VISIT);
      } else {
        k.st = SMS_VISIT;
      }
    }
    // ---------------- visit the chain points in order
    if (k.st == SMS_VISIT && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_VISIT);
      for (;;) {
        if (k.cur >= c.len) {
          k.st = SMS_VIEW_NEXT;
          break;
        }
        if (k.epc_matched && k.cur == k.idx_first) {
          k.cur = k.idx_second;
          k.last_matched = k.idx_second;
          k.cur++;
          continue;
        }
        const ViewCand& vc = c.cand[c.head + k.cur];
        if (!vc.valid) {
          k.cur++;
          continue;
        }
        const uint32_t g = s.view_pl_off[k.v] + vc.pl;
        c.bytes += 8ull * (s.pl_vtx_off[g + 1] - s.pl_vtx_off[g]);
        if (vc.d2 > 16.0f) {  // abandons this view (Q4)
          k.st = SMS_VIEW_NEXT;
          break;
        }
        k.st = SMS_VISIT_ATTACH;
        if (sm_lazy_presolve(s) && c.head + k.cur >= k.spec_slot_hi) {
          int to = k.cur + presolve_window(s);
          if (to > c.len) to = c.len;
          k.gn_from = k.cur;
          k.spec_slot_hi = c.head + to;
          sm_post_gn(k, SMB_PRESOLVE, to - k.cur, SMS_VISIT_ATTACH);
        }
        break;
      }
    }
    if (k.st == SMS_VISIT_ATTACH && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_VISIT_ATTACH);
      const ViewCand& vc = c.cand[c.head + k.cur];
      k.o.view = (uint32_t)k.v;
      k.o.pl = vc.pl;
      k.o.seg = vc.seg;
      k.o.x = vc.x;
      k.o.y = vc.y;
      k.lo = k.last_matched + 1;
      k.ci = k.cur;
      k.hi = k.epc_matched ? (k.cur <= k.idx_first ? k.idx_first : c.len) : c.len;
      k.pre_kind = 2;
      k.ret_to = SMR_VISIT;
      k.st = SMS_ATTACH_BEGIN;
    }
    // ---------------- attach_view
    if (k.st == SMS_ATTACH_BEGIN) {
      const SmProf<Env> prof_(env, SMS_ATTACH_BEGIN);
      k.to_start = 0;
      k.to_end = 0;
      k.attach_ok = 0;
      if (k.pre_kind) {
        const uint32_t ok = k.pre_kind == 1 ? c.epcres[k.e].ok : c.cand[c.head + k.cur].cok;
        const float* X = k.pre_kind == 1 ? c.epcres[k.e].X : c.cand[c.head + k.cur].cX;
        if (!ok) {
          k.st = SMS_ATTACH_RET;
        } else {
          k.X[0] = X[0];
          k.X[1] = X[1];
          k.X[2] = X[2];
          k.st = SMS_ATTACH_SIDES;
        }
      } else {
        sm_post_gn(k, SMB_CENTRAL, 1, SMS_CENTRAL_DONE);
      }
    }
    if (k.st == SMS_CENTRAL_DONE && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_CENTRAL_DONE);
      if (!q.mbox->ok) {
        k.st = SMS_ATTACH_RET;
      } else {
        k.X[0] = q.mbox->X[0];
        k.X[1] = q.mbox->X[1];
        k.X[2] = q.mbox->X[2];
        k.st = SMS_ATTACH_SIDES;
      }
    }
    if (k.st == SMS_ATTACH_SIDES) {
      const SmProf<Env> prof_(env, SMS_ATTACH_SIDES);
      k.nd1 = k.nd2 = 0;
      k.n1 = k.n2 = 0;
      k.which = 0;
      if (k.ci > k.lo) {
        const PlRef pl = polyline_of(s, (int)k.o.view, k.o.pl);
        env.walk_stage(s, c, (int)k.o.view, pl, k.lo, k.ci, k.hi);
        sm_walks(env, s, q, pl, pl.start, pl.end);
        k.which = 1;
        if (k.m1 == 0) {  // orientation A cannot reach the lower neighbour: B at once
          k.which = 2;
          sm_walks(env, s, q, pl, pl.end, pl.start);
        }
        if (k.m1 == 0) {
          k.which = 0;
          k.st = SMS_ATTACH_CHECK;
        } else {
          sm_post_gn(k, SMB_SIDES, k.m1 + k.m2, SMS_SIDES_DONE);
        }
      } else {
        k.st = SMS_ATTACH_CHECK;
      }
    }
    if (k.st == SMS_SIDES_DONE && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_SIDES_DONE);
      const int n1 = sm_leading_ok(c.pend1, k.m1);
      const int n2 = n1 > 0 ? sm_leading_ok(c.pend2, k.m2) : 0;
      k.n1 = n1;
      k.n2 = n2;
      if (n1 > 0) {
        k.st = SMS_ATTACH_CHECK;
      } else if (k.which == 2) {
        k.which = 0;
        k.st = SMS_ATTACH_CHECK;
      } else {
        // A's first start-side solve failed: orientation B on its own
        const PlRef pl = polyline_of(s, (int)k.o.view, k.o.pl);
        env.walk_stage(s, c, (int)k.o.view, pl, k.lo, k.ci, k.hi);
        sm_walks(env, s, q, pl, pl.end, pl.start);
        k.which = 2;
        if (k.m1 == 0) {
          k.which = 0;
          k.st = SMS_ATTACH_CHECK;
        } else {
          sm_post_gn(k, SMB_SIDES, k.m1 + k.m2, SMS_SIDES_DONE);
        }
      }
    }
    if (k.st == SMS_ATTACH_CHECK) {
      const SmProf<Env> prof_(env, SMS_ATTACH_CHECK);
      const int n1 = k.n1, n2 = k.n2, ci = k.ci;
      if ((ci > 0 && n1 == 0) || (ci < c.len - 1 && n2 == 0)) {
        k.st = SMS_ATTACH_RET;
      } else {
        if (k.which) {
          const PlRef pl = polyline_of(s, (int)k.o.view, k.o.pl);
          k.nd1 = k.which == 1 ? pl.start : pl.end;
          k.nd2 = k.which == 1 ? pl.end : pl.start;
        }
        // commit: the 1 + n1 + n2 touched points, in the order of the sequential statement. The points are distinct, so
        // the headers and pending observations of four of them are requested together (one trip to memory per four
        // points instead of one per point: the lane commits alone).
        const int T = 1 + n1 + n2;
        bool overflow = false;
        for (int t0 = 0; t0 < T && !overflow; t0 += 4) {
          ChainPt* pp[4];
          uint32_t hn[4], hc[4], ho[4];
          Obs po[4];
          float pX[4][3];
          for (int b = 0; b < 4; b++) {
            const int tt = t0 + b < T ? t0 + b : t0;
            const Pending* pd = nullptr;
            if (tt == 0)
              pp[b] = &chain_at(c, ci);
            else if (tt <= n1) {
              pp[b] = &chain_at(c, ci - tt);
              pd = &c.pend1[tt - 1];
            } else {
              pp[b] = &chain_at(c, ci + (tt - n1));
              pd = &c.pend2[tt - n1 - 1];
            }
            hn[b] = pp[b]->nobs;
            hc[b] = pp[b]->cap;
            ho[b] = pp[b]->off;
            if (pd) {
              po[b] = pd->o;
              pX[b][0] = pd->X[0];
              pX[b][1] = pd->X[1];
              pX[b][2] = pd->X[2];
            } else {
              po[b] = k.o;
              pX[b][0] = k.X[0];
              pX[b][1] = k.X[1];
              pX[b][2] = k.X[2];
            }
          }
          for (int b = 0; b < 4 && t0 + b < T; b++) {
            ChainPt* p = pp[b];
            uint32_t nobs = hn[b], off = ho[b], need = 0;
            if (nobs == hc[b]) need = hc[b] ? hc[b] * 2 : 4;
            if (c.pool_used + need > c.pool_cap) {
              c.flags |= 2u;  // the host enlarges the pool and reruns the chunk
              overflow = true;
              break;
            }
            if (need) {
              const uint32_t noff = c.pool_used;
              sm_copy_obs(c.pool + noff, c.pool + off, nobs);
              off = noff;
              p->off = noff;
              p->cap = need;
            }
            c.pool[off + nobs] = po[b];
            p->X[0] = pX[b][0];
            p->X[1] = pX[b][1];
            p->X[2] = pX[b][2];
            p->nobs = nobs + 1;
            c.pool_used += need;
          }
        }
        k.to_start = n1;
        k.to_end = n2;
        k.attach_ok = 1;
        k.side = 0;
        k.st = SMS_FOLLOW_SIDE;
      }
    }
    // ---------------- grow the chain at the front, then at the back
    if (k.st == SMS_FOLLOW_END) {
      const SmProf<Env> prof_(env, SMS_FOLLOW_END);
      if (k.side == 0) {
        k.to_start += k.added;
        k.ci += k.added;
      } else {
        k.to_end += k.added;
      }
      k.side++;
      k.st = SMS_FOLLOW_SIDE;
    }
    if (k.st == SMS_FOLLOW_SIDE) {
      const SmProf<Env> prof_(env, SMS_FOLLOW_SIDE);
      for (; k.side < 2; k.side++) {
        const bool front = k.side == 0;
        if (front ? !(k.n1 > 0 && k.n1 == k.ci) : !(k.n2 > 0 && k.n2 == (c.len - k.ci - 1))) continue;
        if (front)
          c.start_dirs[k.o.view] = k.nd1;
        else
          c.end_dirs[k.o.view] = k.nd2;
        k.added = 0;
        k.st = SMS_FOLLOW_STEP;
        break;
      }
      if (k.side >= 2) k.st = SMS_ATTACH_RET;
    }
    if (k.st == SMS_STEP_OK) {
      const SmProf<Env> prof_(env, SMS_STEP_OK);
      const bool front = k.side == 0;
      bool stop = false;
      if (front ? (c.head <= 0) : (c.head + c.len >= c.cap_pts)) {
        c.flags |= 1u;
        stop = true;
      } else {
        ChainPt np;
        if (!new_point_from_tmp(c, np, k.m, k.X)) {
          stop = true;
        } else {
          if (front) {
            c.head--;
            c.pts[c.head] = np;
          } else {
            c.pts[c.head + c.len] = np;
          }
          c.len++;
          k.added++;
        }
      }
      k.st = stop ? SMS_FOLLOW_END : SMS_FOLLOW_STEP;
      if (stop) continue;  // (SMS_FOLLOW_END is above)
    }
    if (k.st == SMS_FOLLOW_STEP) {
      const SmProf<Env> prof_(env, SMS_FOLLOW_STEP);
      k.st_obs = 0;
      k.st = SMS_STEP_CAND;
    }
    // ---------------- N-view step: candidates in observation order
    if (k.st == SMS_STEP_CAND) {
      const SmProf<Env> prof_(env, SMS_STEP_CAND);
      const bool front = k.side == 0;
      const ChainPt& cur = front ? chain_at(c, 0) : chain_at(c, c.len - 1);
      const uint32_t* dirs = front ? c.start_dirs : c.end_dirs;
      const int n = (int)cur.nobs;
      int m = 0;
      while (k.st_obs < n) {
        m = env.step_walks(s, c.pool + cur.off, n, k.st_obs, dirs, c.tmp_a, c.tmp_cap, c.flags);
        if (m) break;
        k.st_obs++;
      }
      if (!m) {
        k.st = SMS_FOLLOW_END;  // no candidate left: the following ends
        continue;
      }
      k.m = m;
      k.st = SMS_STEP_TRI;
    }
    if (k.st == SMS_STEP_TRI) {
      const SmProf<Env> prof_(env, SMS_STEP_TRI);
      sm_list_dlt(env, s, c.tmp_a, k.m, c.flags, k.X);
      sm_post_gn(k, SMB_LIST_A, 1, SMS_STEP_TRI_DONE);
    }
    if (k.st == SMS_STEP_TRI_DONE && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_STEP_TRI_DONE);
      if (q.mbox->ok) {
        k.X[0] = q.mbox->X[0];
        k.X[1] = q.mbox->X[1];
        k.X[2] = q.mbox->X[2];
        k.st = SMS_STEP_OK;
        continue;
      }
      // 3-subset fallback (stepn_fallback)
      if (k.m <= 3) {
        k.st_obs++;
        k.st = SMS_STEP_CAND;
        continue;
      }
      k.fi = 0;
      k.fj = 1;
      k.fk = 2;
      k.st = SMS_FB_NEXT;
    }
    if (k.st == SMS_FB_TRI_DONE && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_FB_TRI_DONE);
      if (q.mbox->ok) {
        k.X[0] = q.mbox->X[0];
        k.X[1] = q.mbox->X[1];
        k.X[2] = q.mbox->X[2];
        for (int i = 0; i < k.m; i++) c.tmp_mask[i] = (i == k.fi || i == k.fj || i == k.fk) ? 1 : 0;
        // (tmp_b[0..3) already holds the three)
        k.kept = 3;
        k.fb_i = 0;
        k.st = SMS_FB_ADD;
      } else {
        // next subset: ascending (i, j, k), k fastest
        k.fk++;
        if (k.fk >= k.m) {
          k.fj++;
          if (k.fj >= k.m - 1) {
            k.fi++;
            k.fj = k.fi + 1;
          }
          k.fk = k.fj + 1;
        }
        k.st = SMS_FB_NEXT;
      }
    }
    if (k.st == SMS_FB_NEXT) {
      const SmProf<Env> prof_(env, SMS_FB_NEXT);
      if (k.fi >= k.m - 2) {  // no valid 3-subset: this candidate is dead
        k.st_obs++;
        k.st = SMS_STEP_CAND;
        continue;
      }
      c.tmp_b[0] = c.tmp_a[k.fi];
      c.tmp_b[1] = c.tmp_a[k.fj];
      c.tmp_b[2] = c.tmp_a[k.fk];
      sm_list_dlt(env, s, c.tmp_b, 3, c.flags, k.X);
      sm_post_gn(k, SMB_LIST_B, 1, SMS_FB_TRI_DONE);
    }
    if (k.st == SMS_FB_ADD_DONE && k.wait == SM_RUN) {
      const SmProf<Env> prof_(env, SMS_FB_ADD_DONE);
      if (q.mbox->ok) {
        c.tmp_mask[k.fb_i] = 1;
        k.X[0] = q.mbox->X[0];
        k.X[1] = q.mbox->X[1];
        k.X[2] = q.mbox->X[2];
        c.tmp_b[k.kept++] = c.tmp_a[k.fb_i];
      }
      k.fb_i++;
      k.st = SMS_FB_ADD;
    }
    if (k.st == SMS_FB_ADD) {
      const SmProf<Env> prof_(env, SMS_FB_ADD);
      while (k.fb_i < k.m && c.tmp_mask[k.fb_i]) k.fb_i++;
      if (k.fb_i < k.m) {
        sm_post_gn(k, SMB_LISTADD, 1, SMS_FB_ADD_DONE);
      } else {
        int kk = 0;
        for (int i = 0; i < k.m; i++)
          if (c.tmp_mask[i]) c.tmp_a[kk++] = c.tmp_a[i];
        k.m = kk;
        k.st = SMS_STEP_OK;
        continue;
      }
    }
    if (k.st == SMS_FINISH) k.wait = SM_DONE;
  }
}

// The finished chain's summary (expand_chain's tail).
EG3D_HD void sm_finish(const SmChain& q, ChainOut& out) {
  const Chain& c = q.c;
  uint32_t nobs = 0;
  for (int i = 0; i < c.len; i++) nobs += c.pts[c.head + i].nobs;
  out.n_points = (uint32_t)c.len;
  out.n_obs = nobs;
  out.flags = c.flags;
  out.head = (uint32_t)c.head;
  out.bytes = c.bytes;
  out.spt = out.sobs = 0;
  for (int k = 0; k < 16; k++) out.tsec[k] = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The side walk of walk_side_candidates_core with its memory trips taken out of the dependent chain (the engine kernel's
// lanes walk alone: nothing else hides a trip). Same tests in the same order on the same operands as the plain walk
// (walk_by_line per chain point) => same candidates, same flags:
//   * the epipolar lines of the next EG3D_SM_EPI_AHEAD chain points are requested together;
//   * the polyline's vertices ahead of the current position live in a window of five (VtxWindow, as walk_by_line_pf)
//     that is carried FROM ONE WALK TO THE NEXT: a walk that ends on segment k of the window leaves the window shifted
//     to that segment, so a side walk opens the window once and every later vertex is requested four tests before it
//     is needed.
#define EG3D_SM_EPI_AHEAD 4
EG3D_HD int sm_side_walk_stream(const DevScene& s, Chain& c, int view, const PlRef& pl, const Obs& from, uint32_t direction,
                                int lo, int ci, int hi, bool towards_start, Pending* out) {
  (void)s;
  const int step_i = towards_start ? -1 : 1;
  int i = towards_start ? ci - 1 : ci + 1;
  int remaining = towards_start ? i - lo + 1 : hi - i;
  if (remaining <= 0) return 0;
  const bool to_start = direction == pl.start;
  const bool bad_dir = !to_start && direction != pl.end;  // Q15: every walk fails (after the epipolar line was checked)
  const int32_t n = (int32_t)pl.n;
  PlPt actual;
  actual.seg = from.seg;
  actual.x = from.x;
  actual.y = from.y;
  VtxWindow<const f2*> W;
  int32_t first = to_start ? (int32_t)actual.seg : (int32_t)actual.seg + 1;
  if (!bad_dir) W.open(pl.v, first, to_start ? -1 : 1, n - 1);
  int cnt = 0;
  while (remaining > 0) {
    // the lines of the next few chain points, requested together
    float eok[EG3D_SM_EPI_AHEAD], ea[EG3D_SM_EPI_AHEAD], eb[EG3D_SM_EPI_AHEAD], ec[EG3D_SM_EPI_AHEAD];
    const int nb = remaining < EG3D_SM_EPI_AHEAD ? remaining : EG3D_SM_EPI_AHEAD;
    for (int b = 0; b < EG3D_SM_EPI_AHEAD; b++) {
      const int ib = b < nb ? i + step_i * b : i;
      const ViewCand& ve = c.cand[c.head + ib];
      eok[b] = ve.eok ? 1.0f : 0.0f;
      ea[b] = ve.ea;
      eb[b] = ve.eb;
      ec[b] = ve.ec;
    }
    for (int b = 0; b < nb; b++) {
      if (eok[b] == 0.0f) return cnt;
      if (bad_dir) {
        c.flags |= 8u;
        return cnt;
      }
      const float la = ea[b], lb = eb[b], lc = ec[b];
      // ---- next hit of the line from `actual` towards `direction` (walk_by_line_pf on the carried window)
      const LineDir ld = line_dir(la, lb);
      const int32_t count = to_start ? first + 1 : n - first;  // vertices ahead: u[0] = v[first], ...
      float hx = 0.0f, hy = 0.0f;
      uint32_t seg_found = actual.seg;
      uint32_t r = seg_line_hit_guarded(actual.x, actual.y, W.w0.x, W.w0.y, la, lb, lc, ld, hx, hy);
      if (r & 2u) return cnt;  // quasi-parallel: the side walk ends
      if (!(r & 1u)) {
        bool got = false;
        int32_t kk = 1;
        for (; kk < count; kk++) {
          r = seg_line_hit_guarded(W.w0.x, W.w0.y, W.w1.x, W.w1.y, la, lb, lc, ld, hx, hy);
          if (r & 2u) return cnt;
          if (r & 1u) {
            got = true;
            seg_found = (uint32_t)(to_start ? first - kk : first + kk - 1);
            break;
          }
          W.shift(kk);
        }
        if (!got) return cnt;  // reached the extreme
        // the next walk starts on the segment of the hit: its u[0] is this walk's u[kk]
        W.shift(kk);
        first += W.stepv * kk;
        W.first = first;
      }
      Pending& pd = out[cnt++];
      pd.o.view = (uint32_t)view;
      pd.o.pl = from.pl;
      pd.o.seg = seg_found;
      pd.o.x = hx;
      pd.o.y = hy;
      pd.ok = 0;
      actual.seg = seg_found;
      actual.x = hx;
      actual.y = hy;
    }
    i += step_i * nb;
    remaining -= nb;
  }
  return cnt;
}

// The walk phase of one candidate of the N-view step (stepn_walks with one member) with its memory trips batched: the
// observations are taken four at a time and every stage of the four — the observations themselves; their epipolar lines,
// directions and polyline indices; the polyline descriptors; the first vertices ahead — is requested together before
// the next stage needs it. Same walks (walk_by_distance_pf / walk_by_line_pf: the tests of the plain walks in the same
// order, tests/test_cpu_parity.py) in observation order => same list, same flags.
#define EG3D_SM_STEP_CHUNK 4
EG3D_HD int sm_step_walks_stream(const DevScene& s, const Obs* co_all, int n, int st, const uint32_t* dirs, Obs* sel,
                                 int sel_cap, uint32_t& flags) {
  const Obs so = co_all[st];
  const PlRef ps = polyline_of(s, (int)so.view, so.pl);
  PlPt p, q;
  p.seg = so.seg;
  p.x = so.x;
  p.y = so.y;
  const uint32_t w = walk_by_distance_pf(ps, p, dirs[so.view], EG3D_FOLLOW_STEP, q);
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
  bool full = false;
  constexpr int CH = EG3D_SM_STEP_CHUNK;
  for (int i0 = 0; i0 < n && !full; i0 += CH) {
    Obs co[CH];
    bool use[CH], eok[CH], open[CH];
    float la[CH], lb[CH], lc[CH];
    uint32_t dir[CH], g[CH];
    PlRef pk[CH];
    VtxWindow<const f2*> W[CH];
    for (int b = 0; b < CH; b++) {
      const int i = i0 + b;
      use[b] = i < n && i != st;
      co[b] = co_all[use[b] ? i : st];
    }
    for (int b = 0; b < CH; b++) {
      eok[b] = epiline(s.F, s.F_valid, s.n_views, (int)so.view, (int)co[b].view, q.x, q.y, la[b], lb[b], lc[b]) && use[b];
      dir[b] = dirs[co[b].view];
      g[b] = s.view_pl_off[co[b].view] + co[b].pl;
    }
    for (int b = 0; b < CH; b++) {
      const uint32_t va = s.pl_vtx_off[g[b]], vb = s.pl_vtx_off[g[b] + 1];
      pk[b].v = s.vtx + va;
      pk[b].n = vb - va;
      pk[b].start = s.pl_start[g[b]];
      pk[b].end = s.pl_end[g[b]];
    }
    for (int b = 0; b < CH; b++) {
      PlPt cp;
      cp.seg = co[b].seg;
      cp.x = co[b].x;
      cp.y = co[b].y;
      open[b] = walk_by_line_open(pk[b], cp, dir[b], W[b]);
    }
    for (int b = 0; b < CH; b++) {
      if (!eok[b]) continue;
      if (!open[b]) {
        fl |= 8u;  // WALK_BAD_DIR
        continue;
      }
      PlPt cp, rp;
      cp.seg = co[b].seg;
      cp.x = co[b].x;
      cp.y = co[b].y;
      const uint32_t wr = walk_by_line_run(pk[b], cp, dir[b], la[b], lb[b], lc[b], true, EG3D_FOLLOW_MIN, EG3D_FOLLOW_MAX, W[b], rp);
      if (wr & WALK_FOUND) {
        if (m < sel_cap) {
          sel[m].view = co[b].view;
          sel[m].pl = co[b].pl;
          sel[m].seg = rp.seg;
          sel[m].x = rp.x;
          sel[m].y = rp.y;
        }
        if (m + 1 > sel_cap) {
          flags |= 2u;
          m = sel_cap;
          full = true;
          break;
        }
        m++;
      }
    }
  }
  flags |= fl;
  return m < 3 ? 0 : m;
}

// Lane-private primitives of the machine in their plain (sequential) form: what the host simulation uses, and the
// engine kernel unless it overrides one.
struct SmEnvSeq {
  EG3D_HD void prof_begin(uint32_t) const {}
  EG3D_HD void prof_end(uint32_t) const {}
  EG3D_HD void dlt(const float* P1, float x1, float y1, const float* P2, float x2, float y2, double X0[3]) const {
    dlt2(P1, x1, y1, P2, x2, y2, X0);
  }
  EG3D_HD void walk_stage(const DevScene&, Chain&, int, const PlRef&, int, int, int) const {}
  EG3D_HD int side_walk(const DevScene& s, Chain& c, int view, const PlRef& pl, const Obs& from, uint32_t direction, int lo,
                        int ci, int hi, bool towards_start, Pending* out) const {
    return TeamSeq().side_walk(s, c, view, pl, from, direction, lo, ci, hi, towards_start, out);
  }
  EG3D_HD int step_walks(const DevScene& s, const Obs* co_all, int n, int st, const uint32_t* dirs, Obs* sel, int sel_cap,
                         uint32_t& flags) const {
    return stepn_walks(TeamSeq(), s, co_all, n, st, dirs, sel, sel_cap, flags);
  }
};

// ... with the memory trips of the lane-private loops taken out of the dependent chains (what the engine kernel runs;
// hostsim mode 3 runs it on the CPU against the oracle)
struct SmEnvStream : SmEnvSeq {
  // (the 2-view DLT stays in registers, SmEnvSeq::dlt: with its matrices in lane-private memory — dlt2_mem on a scratch
  // array, which frees ~80 registers — a decomposition took 150 k clocks, 37 % of the engine's advance phase:
  // profiles/r05_experiments/engine_blocks_c3.txt)
  EG3D_HD int side_walk(const DevScene& s, Chain& c, int view, const PlRef& pl, const Obs& from, uint32_t direction, int lo,
                        int ci, int hi, bool towards_start, Pending* out) const {
    return sm_side_walk_stream(s, c, view, pl, from, direction, lo, ci, hi, towards_start, out);
  }
  EG3D_HD int step_walks(const DevScene& s, const Obs* co_all, int n, int st, const uint32_t* dirs, Obs* sel, int sel_cap,
                         uint32_t& flags) const {
    return sm_step_walks_stream(s, co_all, n, st, dirs, sel, sel_cap, flags);
  }
};

}  // namespace eg3d
