// eg3d_k3a_engine.h — K3a as a request/serve engine (gfx950 only; included by eg3d_kernels.hip).
//
// What the hypothesis stage computes is defined by eg3d_dev_follow.h (evaluate_hypothesis with HTeamSeq is the
// sequential statement the host simulation runs; reference: plg_matching.cpp:51-265, 633-795, 1276-1287). This file
// is how the GPU runs it. The cost of a hypothesis is its TRIANGULATIONS (2-view DLT by a Jacobi SVD + FP64
// Gauss-Newton on three rows: ~5000 dependent instructions on one lane); the walks between them are cheap and
// irregular. One lane per hypothesis with the triangulation inlined wherever the control flow needs one ran the
// wave at 10-16 % active lanes (profiles/r03_c3_rocprof_summary.txt: every lane waits at its own call site while
// the others walk, fail, or have finished). Here a wavefront is a small server:
//
//   every iteration   (1) each lane advances its own state machine — evaluate the answers it got, fetch the next
//                         hypothesis / list, walk — up to its next triangulation REQUESTS (three observations each),
//                     (2) the requests go into the wave's 64 slots in LDS (lanes that are still working are ranked;
//                         each gets K = 64 / #working slots, a power of two, at most EG3D_K3A_SPEC),
//                     (3) ALL 64 lanes serve: lane s triangulates slot s — one call site, dense,
//                     (4) owners read their results back.
//
// Because K grows as lanes run out of work, the tail of a launch parallelises by itself: the 4 direction
// combinations of an orientation round are served at once, a replay walks K steps ahead (its steps are known to
// succeed), and a following list SPECULATES K steps ahead (walks depend on earlier walks only; the triangulations
// of steps i+1.. are served together with step i's and dropped if it fails). Results — points, flags, order — are
// those of the sequential statement: a speculative request's flags are committed only when everything before it
// succeeded.
//
// The orientation search also stops repeating itself: in a lock-step round the four (dirB, dirC) combinations share
// the +10 px walk on A, the two epipolar lines, and each of the two walks on B and on C (a combination's walk on C
// is made exactly when its walk on B found something, as in step3) — 1+2+2 walks per round instead of 4+4+4; the
// positions are kept per polyline direction (A, B[2], C[2]), in place, in LDS.
#pragma once
#include "eg3d_dev_pipeline.h"

#include "eg3d_dev_coopgn.h"

namespace eg3d {

// timing build: shader clocks of a wave's phases, summed over the launch (g_gn_dbg[113 + 7*kernel ..]: advance, serve,
// read answers; iterations, requests served, working lanes) — tools/k3a_stats.py
#ifdef EG3D_GN_COUNTERS
#define K3A_T0() unsigned long long kt_[3] = {0, 0, 0}, kc_[4] = {0, 0, 0, 0}, kt0_ = __builtin_readcyclecounter(), kt1_
#define K3A_T(i) (kt1_ = __builtin_readcyclecounter(), kt_[i] += kt1_ - kt0_, kt0_ = kt1_)
#define K3A_C(i, v) (kc_[i] += (v))
#define K3A_TEND(kern)                                                \
  if (lane == 0) {                                                    \
    for (int q_ = 0; q_ < 3; q_++) EG3D_GN_DBG(113 + 7 * (kern) + q_, kt_[q_]); \
    for (int q_ = 0; q_ < 4; q_++) EG3D_GN_DBG(116 + 7 * (kern) + q_, kc_[q_]); \
  }
#else
#define K3A_T0() ((void)0)
#define K3A_T(i) ((void)0)
#define K3A_C(i, v) ((void)0)
#define K3A_TEND(kern) ((void)0)
#endif

#define EG3D_K3A_STAGE EG3D_K3A_STAGE_POINTS /* followed points staged per lane before a list moves into the arena (eg3d_kernels.h) */
#ifndef EG3D_K3A_SPEC
#define EG3D_K3A_SPEC 32 /* most requests one lane may issue per iteration (look-ahead depth): 4 / 8 / 16 / 32 / 64 -> C3' K3a 5.36 / 4.74 / 4.58 / 4.43 / 4.47 ms on one box (other boxes: +-0.2 ms) */
#endif

struct K3aShared {
  Obs req[64][3];          // request slots
  float resX[64][3];       // served: the point
  uint32_t resF[64];       // served: bit 0 = valid point, bits 8.. = EG3D_FLAG_* raised by the triangulation
  uint32_t reqM[64];       // issued: bit 0 = slot holds a request; follow: bits 8-9 = starting observation, 16.. = walk flags
  uint32_t pos[6][3][64];  // per-lane positions (seg, x, y): orientation 0 = A, 1-2 = B[dir], 3-4 = C[dir];
                           // following 0-2 = the list's last point, 3-5 = the base the next request is generated from
  uint32_t plr[3][4][64];  // per-lane polylines of the hypothesis' three views: first vertex, vertex count, start / end node
};
static_assert(sizeof(K3aShared) <= 12800, "K3aShared must fit 10 LDS allocation units (12 single-wave blocks per CU)");

struct K3aLane {
  K3aShared* sh;
  uint32_t lane;
  __device__ __forceinline__ PlPt get(int e) const {
    PlPt p;
    p.seg = sh->pos[e][0][lane];
    p.x = __uint_as_float(sh->pos[e][1][lane]);
    p.y = __uint_as_float(sh->pos[e][2][lane]);
    return p;
  }
  __device__ __forceinline__ void set(int e, const PlPt& p) const {
    sh->pos[e][0][lane] = p.seg;
    sh->pos[e][1][lane] = __float_as_uint(p.x);
    sh->pos[e][2][lane] = __float_as_uint(p.y);
  }
  // the polyline of slot k, looked up once per hypothesis / list (polyline_of is two dependent trips to memory)
  __device__ __forceinline__ void load_polyline(const DevScene& s, int k, int view, uint32_t pl_id) const {
    const uint32_t g = s.view_pl_off[view] + pl_id;
    const uint32_t a = s.pl_vtx_off[g], b = s.pl_vtx_off[g + 1];
    sh->plr[k][0][lane] = a;
    sh->plr[k][1][lane] = b - a;
    sh->plr[k][2][lane] = s.pl_start[g];
    sh->plr[k][3][lane] = s.pl_end[g];
  }
  __device__ __forceinline__ PlRef polyline(const DevScene& s, int k) const {
    PlRef r;
    r.v = s.vtx + sh->plr[k][0][lane];
    r.n = sh->plr[k][1][lane];
    r.start = sh->plr[k][2][lane];
    r.end = sh->plr[k][3][lane];
    return r;
  }
};

// slots per working lane: the largest power of two <= 64 / n_working, at most EG3D_K3A_SPEC
__device__ __forceinline__ uint32_t k3a_slots_per_lane(uint32_t n_working) {
  uint32_t k = 64u / n_working, p = 1;
  while (p * 2 <= k && p * 2 <= (uint32_t)EG3D_K3A_SPEC) p *= 2;
  return p;
}

// (3) of the loop: every lane serves the slot of its own index
__device__ __forceinline__ void k3a_serve(K3aShared& sh, const DevScene& s, uint32_t lane) {
  __syncthreads();
  if (sh.reqM[lane] & 1u) {
    const Obs o[3] = {sh.req[lane][0], sh.req[lane][1], sh.req[lane][2]};
    float X[3] = {0.0f, 0.0f, 0.0f};
    uint32_t f = 0;
    const bool ok = triangulate3(s.cam_P, o, X, f);
    sh.resX[lane][0] = X[0];
    sh.resX[lane][1] = X[1];
    sh.resX[lane][2] = X[2];
    sh.resF[lane] = (ok ? 1u : 0u) | (f << 8);
  }
  __syncthreads();
}

__device__ __forceinline__ HPoint k3a_point_of_slot(const K3aShared& sh, uint32_t slot) {
  HPoint hp;
  hp.X[0] = sh.resX[slot][0];
  hp.X[1] = sh.resX[slot][1];
  hp.X[2] = sh.resX[slot][2];
  hp.nobs = 3;
  hp.o[0] = sh.req[slot][0];
  hp.o[1] = sh.req[slot][1];
  hp.o[2] = sh.req[slot][2];
  hp.pad = 0;
  return hp;
}

// ---------------------------------------------------------------- orientation --------
// One lane per hypothesis: triangulate the three hits, orientation search from A's first extreme (and from the
// other one if that fails), replay of the surviving combination's points into the arena, the single step in the
// opposite direction. Leaves in res[h] what k3a_follow_spec extends — status (TRI, D1, D2), n1 / n2 and the offsets
// of the initial lists, the directions, the central point, the flags — and appends the lists to follow to `items`.
//
// A lane's state machine is run as a PIPELINE, each stage once per iteration and in this order, so that the wave
// executes every stage's code once however its lanes are spread over the states (a free-running loop executed the
// walks up to five times per iteration: 77 % of the wave's time): END (evaluate the answers of a round: next round,
// other extreme, replay, opposite step, or write the result) -> IDLE (take the next hypothesis) -> GEN (walk a round
// and issue its requests; with slots to spare walk and issue the next rounds too) -> REQ (requests not yet issued).
// Looking ahead is safe in the search: positions are kept per polyline direction and never depend on which
// combinations are alive, a dead combination's later answers are ignored, and the two flags a step can raise
// (direction mismatch, degenerate DLT) depend on the polylines, directions and views only — a looked-ahead round can
// only repeat flags its first round raised.
enum : uint32_t { K3A_IDLE = 0, K3A_GEN = 1, K3A_REQ = 2, K3A_END = 3 };
enum : uint32_t { K3A_TRI0 = 0, K3A_ORIENT = 1, K3A_REPLAY = 2, K3A_OPP = 3 };

__global__ void __launch_bounds__(64, EG3D_K3A_WAVES) k3a_orient(DevScene s, StageAView a, const TaskDesc* tasks,
                                                                 const uint32_t* hyp_off, uint32_t n_hyp, HypResult* res,
                                                                 uint32_t cap, HPoint* arena, uint32_t arena_cap,
                                                                 Counters* ctr, uint32_t* queue, uint32_t lanes_per_wave,
                                                                 uint32_t* items, uint32_t* n_items) {
  __shared__ K3aShared sh;
  const uint32_t lane = threadIdx.x;
  const K3aLane L{&sh, lane};
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t st = K3A_IDLE, phase = K3A_TRI0;
  bool exhausted = lane >= lanes_per_wave;  // small batches: fewer working lanes per wave = more slots for each
  uint32_t h = 0, t_cur = 0, e = 0, rounds = 0, alive = 0, reqmask = 0, okmask = 0, flags = 0, status = 0;
  uint32_t n1 = 0, n2 = 0, i_rep = 0, base = 0xffffffffu;
  uint32_t dead_round = 0xffffffffu;  // a looked-ahead round in which no combination found its walks
  bool arena_ok = false;
  int32_t view[3] = {0, 0, 0};
  uint32_t pl[3] = {0, 0, 0};
  uint32_t dirA = 0, dirB[2] = {0, 0}, dirC[2] = {0, 0};
  uint32_t d1[3] = {0, 0, 0}, d2[3] = {0, 0, 0};
  float X0[3] = {0.0f, 0.0f, 0.0f};
  // positions back to the three hits (A; both directions of B and of C)
  auto to_hits = [&](const Obs c[3]) {
    PlPt p;
    p.seg = c[0].seg;
    p.x = c[0].x;
    p.y = c[0].y;
    L.set(0, p);
    p.seg = c[1].seg;
    p.x = c[1].x;
    p.y = c[1].y;
    L.set(1, p);
    L.set(2, p);
    p.seg = c[2].seg;
    p.x = c[2].x;
    p.y = c[2].y;
    L.set(3, p);
    L.set(4, p);
  };
  auto reload_hits = [&]() {  // (restarts are rare: the hits are fetched again instead of being kept)
    Obs c[3];
    hypothesis_hits(a, tasks[t_cur], t_cur, h - hyp_off[t_cur], c);
    to_hits(c);
  };
  // request of combination q = 2*b + c from the current positions, tagged with the round it belongs to
  auto issue = [&](uint32_t slot, int q, uint32_t round_off) {
    const PlPt pA = L.get(0), pB = L.get(1 + (q >> 1)), pC = L.get(3 + (q & 1));
    Obs o;
    o.view = (uint32_t)view[0];
    o.pl = pl[0];
    o.seg = pA.seg;
    o.x = pA.x;
    o.y = pA.y;
    sh.req[slot][0] = o;
    o.view = (uint32_t)view[1];
    o.pl = pl[1];
    o.seg = pB.seg;
    o.x = pB.x;
    o.y = pB.y;
    sh.req[slot][1] = o;
    o.view = (uint32_t)view[2];
    o.pl = pl[2];
    o.seg = pC.seg;
    o.x = pC.x;
    o.y = pC.y;
    sh.req[slot][2] = o;
    sh.reqM[slot] = 1u | ((uint32_t)q << 8) | (round_off << 10);
  };
  K3A_T0();

  for (;;) {
    const unsigned long long working = __ballot(!(st == K3A_IDLE && exhausted));
    if (!working) break;
    K3A_C(0, 1);
    K3A_C(2, __popcll(working));
    const uint32_t K = k3a_slots_per_lane((uint32_t)__popcll(working));
    const uint32_t slot0 = (uint32_t)__popcll(working & lt) * K;
    sh.reqM[lane] = 0;
    __syncthreads();
    uint32_t n_issued = 0;
    // ---- END: every request of the round has been answered (okmask), or the round had none
    if (st == K3A_END) {
      bool finish = false, to_d1 = false, restart = false;
      if (phase == K3A_TRI0) {
        if (okmask & 1u) {
          status |= HYP_TRI;
          e = 0;
          phase = K3A_ORIENT;
          restart = true;
        } else {
          finish = true;
        }
      } else if (phase == K3A_ORIENT) {
        // (alive / rounds were brought up to date by the consumer)
        const int amount = __popc(alive);
        if (amount > 1) {
          st = K3A_GEN;
        } else if (amount == 1) {
          to_d1 = true;
        } else if (e == 0) {
          e = 1;
          reload_hits();
          restart = true;
        } else {
          finish = true;
        }
      } else if (phase == K3A_REPLAY) {
        if (i_rep < n1) {
          st = K3A_GEN;
        } else if (e == 0) {
          // the opposite direction is tested once, and only when A was followed towards its start
          phase = K3A_OPP;
          dirA = d2[0];
          dirB[0] = d2[1];
          dirC[0] = d2[2];
          alive = 1u;
          reload_hits();
          st = K3A_GEN;
        } else {
          finish = true;
        }
      } else {  // K3A_OPP
        finish = true;
      }
      if (restart) {  // the search from extreme e
        rounds = 0;
        alive = 15u;
        dirA = sh.plr[0][e == 0 ? 2 : 3][lane];
        dirB[0] = sh.plr[1][2][lane];
        dirB[1] = sh.plr[1][3][lane];
        dirC[0] = sh.plr[2][2][lane];
        dirC[1] = sh.plr[2][3][lane];
        st = K3A_GEN;
      }
      if (to_d1) {
        const int qs = __ffs((int)alive) - 1;
        const PlRef pa = L.polyline(s, 0), pb = L.polyline(s, 1), pc = L.polyline(s, 2);
        status |= HYP_D1;
        d1[0] = dirA;
        d1[1] = dirB[qs >> 1];
        d1[2] = dirC[qs & 1];
        d2[0] = (pa.start == d1[0]) ? pa.end : pa.start;
        d2[1] = (pb.start == d1[1]) ? pb.end : pb.start;
        d2[2] = (pc.start == d1[2]) ? pc.end : pc.start;
        if (rounds > cap) flags |= 4u;
        n1 = rounds < cap ? rounds : cap;
        const uint32_t need = n1 + (e == 0 ? 1u : 0u);
        base = atomicAdd(&ctr->arena_used, need);
        arena_ok = base + need <= arena_cap;
        if (!arena_ok) atomicOr(&ctr->flags, CTR_ARENA_OVERFLOW);
        // replay: the surviving combination alone, from the hits
        phase = K3A_REPLAY;
        i_rep = 0;
        dirA = d1[0];
        dirB[0] = d1[1];
        dirC[0] = d1[2];
        alive = 1u;
        reload_hits();
        st = K3A_GEN;  // (n1 >= 1: the survivor made at least the round that singled it out)
      }
      if (finish) {
        HypResult r;
        r.status = status;
        r.n1 = n1;
        r.n2 = n2;
        for (int k = 0; k < 3; k++) {
          r.dirs1[k] = d1[k];
          r.dirs2[k] = d2[k];
          r.X[k] = X0[k];
        }
        r.flags = flags;
        r.pts1_off = ((status & HYP_D1) && arena_ok) ? base : 0xffffffffu;
        r.pts2_off = ((status & HYP_D2) && arena_ok) ? base + n1 : 0xffffffffu;
        res[h] = r;
        if (flags) atomicOr(&ctr->flags, flags);
        if ((status & HYP_D1) && arena_ok) {  // the lists k3a_follow_spec extends
          const uint32_t cnt = (status & HYP_D2) ? 2u : 1u;
          const uint32_t at = atomicAdd(n_items, cnt);
          items[at] = 2u * h;
          if (cnt == 2u) items[at + 1] = 2u * h + 1u;
        }
        st = K3A_IDLE;
      }
    }
    // ---- IDLE: the next hypothesis; its first request is the triangulation of the hits themselves
    if (st == K3A_IDLE && !exhausted) {
      const uint32_t i = atomicAdd(queue, 1u);
      if (i >= n_hyp) {
        exhausted = true;
      } else {
        h = i;
        t_cur = find_owner(hyp_off, a.n_tasks, h);
        const TaskDesc d = tasks[t_cur];
        Obs c[3];
        hypothesis_hits(a, d, t_cur, h - hyp_off[t_cur], c);
        for (int k = 0; k < 3; k++) {
          view[k] = (int32_t)c[k].view;
          pl[k] = c[k].pl;
          L.load_polyline(s, k, view[k], pl[k]);
        }
        to_hits(c);
        status = 0;
        flags = 0;
        n1 = n2 = 0;
        base = 0xffffffffu;
        arena_ok = false;
        for (int k = 0; k < 3; k++) d1[k] = d2[k] = 0;
        X0[0] = X0[1] = X0[2] = 0.0f;
        phase = K3A_TRI0;
        reqmask = 1u;
        okmask = 0;
        st = K3A_REQ;
      }
    }
    // ---- GEN: walk a round and issue its requests; with slots to spare, the rounds behind it too
    dead_round = 0xffffffffu;
    if (st == K3A_GEN) {
      uint32_t walkers = alive;  // the combinations this round is walked for
      for (uint32_t round_off = 0;; round_off++) {
        // +10 px on A; next epipolar hit (unbounded) on B / C for every direction a live combination uses
        // (combination q = 2*b + c; the walk on C is made for the combinations whose walk on B found something)
        uint32_t fl = 0, m = walkers;
        const PlRef pa = L.polyline(s, 0);
        PlPt q;
        const uint32_t w = walk_by_distance_pf(pa, L.get(0), dirA, EG3D_FOLLOW_STEP, q);
        if (w & WALK_BAD_DIR) fl |= 8u;
        if (w & WALK_EXTREME) {
          m = 0;
        } else {
          L.set(0, q);
          float la = 0.0f, lb = 0.0f, lc = 0.0f;
          PlRef pk = pa;
#pragma unroll 1
          for (int idx = 0; idx < 4 && m; idx++) {
            const int k = 1 + (idx >> 1), j = idx & 1;
            const uint32_t users = k == 1 ? (3u << (2 * j)) : (5u << j);  // the combinations that walk (k, j)
            if (j == 0) {
              if (!epiline(s.F, s.F_valid, s.n_views, view[0], view[k], q.x, q.y, la, lb, lc)) {
                m = 0;  // step3 fails here for every combination
                break;
              }
              pk = L.polyline(s, k);
            }
            if (!(m & users)) continue;
            PlPt r;
            const uint32_t wr = walk_by_line_pf(pk, L.get(2 * k - 1 + j), k == 1 ? dirB[j] : dirC[j], la, lb, lc, false, 0.0f,
                                                0.0f, r);
            if (wr & WALK_BAD_DIR) fl |= 8u;
            if (wr & WALK_FOUND)
              L.set(2 * k - 1 + j, r);
            else
              m &= ~users;
          }
        }
        if (phase != K3A_REPLAY) flags |= fl;  // (the replay repeats steps whose flags are already counted)
        if (round_off == 0) {
          reqmask = m;
          okmask = 0;
          if (!m) {
            if (phase == K3A_REPLAY) n1 = i_rep;  // (a replayed walk found nothing: cannot happen; the list ends there)
            if (phase == K3A_ORIENT) {
              alive = 0;
              rounds++;
            }
            st = K3A_END;
            break;
          }
          st = K3A_REQ;
          while (reqmask && n_issued < K) {
            const int qq = __ffs((int)reqmask) - 1;
            reqmask &= reqmask - 1u;
            issue(slot0 + n_issued, qq, 0u);
            n_issued++;
          }
          if (reqmask) break;  // the rest of this round next iteration
        } else {
          if (!m) {
            if (phase == K3A_REPLAY)
              n1 = i_rep + round_off;
            else
              dead_round = round_off;
            break;
          }
          uint32_t mm = m;
          while (mm) {  // (fits: the look-ahead below checked the room)
            const int qq = __ffs((int)mm) - 1;
            mm &= mm - 1u;
            issue(slot0 + n_issued, qq, round_off);
            n_issued++;
          }
        }
        // look ahead? a replay: its next step (known to succeed); the search: the next round of the combinations
        // that are still walking, if there are at least two and their requests fit
        const uint32_t cm = (uint32_t)__popc(m);
        bool more;
        if (phase == K3A_REPLAY)
          more = i_rep + round_off + 1u < n1 && n_issued < K;
        else if (phase == K3A_ORIENT)
          more = cm >= 2u && n_issued + cm <= K && round_off < 14u;
        else
          more = false;
        if (!more) break;
        walkers = m;
      }
    }
    // ---- REQ: requests of the round that are still to be issued (the first one of a new hypothesis; what did not fit)
    if (st == K3A_REQ && n_issued == 0) {
      while (reqmask && n_issued < K) {
        const int qq = __ffs((int)reqmask) - 1;
        reqmask &= reqmask - 1u;
        issue(slot0 + n_issued, qq, 0u);
        n_issued++;
      }
    }
    // ---- serve
    K3A_T(0);
    K3A_C(1, __popcll(__ballot(sh.reqM[lane] & 1u)));
    k3a_serve(sh, s, lane);
    K3A_T(1);
    // ---- the answers
    if (n_issued) {
      if (phase == K3A_REPLAY) {
        for (uint32_t j = 0; j < n_issued; j++)
          if (arena_ok) arena[base + i_rep + j] = k3a_point_of_slot(sh, slot0 + j);
        i_rep += n_issued;
        st = K3A_END;
      } else if (phase == K3A_ORIENT) {
        // rounds in order; a round is closed when the next one's answers begin (or at the end, if it was issued whole)
        uint32_t cur = 0;
        bool decided = false;
        for (uint32_t j = 0; j < n_issued && !decided; j++) {
          const uint32_t mq = sh.reqM[slot0 + j], f = sh.resF[slot0 + j];
          const uint32_t q = (mq >> 8) & 3u, ro = mq >> 10;
          if (ro != cur) {
            alive = cur == 0 ? okmask : (alive & okmask);
            okmask = 0;
            rounds++;
            cur = ro;
            if (__popc(alive) <= 1) {
              decided = true;
              break;
            }
          }
          if (cur != 0 && !((alive >> q) & 1u)) continue;  // this combination died in an earlier round
          flags |= f >> 8;
          if (f & 1u) okmask |= 1u << q;
        }
        if (decided) {
          st = K3A_END;
        } else if (reqmask) {
          st = K3A_REQ;  // round 0 is not complete yet
        } else {
          alive = cur == 0 ? okmask : (alive & okmask);
          okmask = 0;
          rounds++;
          if (__popc(alive) > 1 && dead_round != 0xffffffffu) {  // the round behind the last issued one had no walk left
            alive = 0;
            rounds++;
          }
          st = K3A_END;
        }
      } else {  // K3A_TRI0, K3A_OPP: one request
        const uint32_t f = sh.resF[slot0];
        flags |= f >> 8;
        if (f & 1u) {
          okmask |= 1u;
          if (phase == K3A_TRI0) {
            X0[0] = sh.resX[slot0][0];
            X0[1] = sh.resX[slot0][1];
            X0[2] = sh.resX[slot0][2];
          } else {
            status |= HYP_D2;
            n2 = 1;
            if (arena_ok) arena[base + n1] = k3a_point_of_slot(sh, slot0);
          }
        }
        st = K3A_END;
      }
    }
    K3A_T(2);
  }
  K3A_TEND(0);
}

// ---------------------------------------------------------------- following --------
// The lists of all hypotheses, one item = (hypothesis, direction), pulled from a queue by whichever lane is free.
// A step of a list (stepn3: try each observation as the +10 px one, bounded epipolar walks on the other two,
// triangulate the first complete triple, next observation if that fails) becomes: generate the next request from
// (last point, first observation still to try), serve, and on failure resume behind the observation that was
// tried. With spare slots the lane generates the following steps' requests too, each from the previous request's
// observations, as if every triangulation before it succeeded.
template <class T>
__device__ __forceinline__ T k3a_pick(const T a[3], int i) {
  return i == 0 ? a[0] : i == 1 ? a[1] : a[2];
}
// next request of a step: the base point's observations are entries 3..5 of the lane's positions, their views in
// view3 (their polylines: the lane's cached polyline of that view). Returns the starting observation used and the
// three new positions in the order the new point lists them (the starting observation, then the other two in index
// order).
__device__ __forceinline__ bool k3a_follow_request(const DevScene& s, const K3aLane& L, const int32_t view3[3],
                                                   int st_start, const uint32_t dirs[3], const int32_t ids[3],
                                                   int& st_used, PlPt sel_pt[3], uint32_t& fl) {
#pragma unroll 1
  for (int st = st_start; st < 3; st++) {
    const int32_t sv = k3a_pick(view3, st);
    const int sd = (sv == ids[0]) ? 0 : (sv == ids[1]) ? 1 : 2;
    const PlRef ps = L.polyline(s, sd);
    PlPt q;
    const uint32_t w = walk_by_distance_pf(ps, L.get(3 + st), k3a_pick(dirs, sd), EG3D_FOLLOW_STEP, q);
    if (w & WALK_BAD_DIR) fl |= 8u;
    if (w & WALK_EXTREME) continue;
    int found = 0;
    PlPt r1 = q, r2 = q;
#pragma unroll 1
    for (int t = 0; t < 2; t++) {
      const int i = t == 0 ? (st == 0 ? 1 : 0) : (st == 2 ? 1 : 2);
      const int32_t cv = k3a_pick(view3, i);
      float la, lb, lc;
      if (!epiline(s.F, s.F_valid, s.n_views, sv, cv, q.x, q.y, la, lb, lc)) continue;
      const int cd = (cv == ids[0]) ? 0 : (cv == ids[1]) ? 1 : 2;
      const PlRef pk = L.polyline(s, cd);
      PlPt r;
      const uint32_t wr =
          walk_by_line_pf(pk, L.get(3 + i), k3a_pick(dirs, cd), la, lb, lc, true, EG3D_FOLLOW_MIN, EG3D_FOLLOW_MAX, r);
      if (wr & WALK_BAD_DIR) fl |= 8u;
      if (wr & WALK_FOUND) {
        found++;
        if (t == 0)
          r1 = r;
        else
          r2 = r;
      }
    }
    if (found < 2) continue;
    st_used = st;
    sel_pt[0] = q;
    sel_pt[1] = r1;
    sel_pt[2] = r2;
    return true;
  }
  return false;
}

__global__ void __launch_bounds__(64, EG3D_K3A_WAVES) k3a_follow_spec(DevScene s, HypResult* res, HPoint* scratch,
                                                                      uint32_t cap, HPoint* arena, uint32_t arena_cap,
                                                                      Counters* ctr, uint32_t* queue, uint32_t lanes_per_wave,
                                                                      const uint32_t* items, const uint32_t* n_items_p) {
  __shared__ K3aShared sh;
  const uint32_t lane = threadIdx.x;
  const K3aLane L{&sh, lane};
  const unsigned long long lt = (1ull << lane) - 1ull;
  // new points of the current list: the first `stage` are staged per lane (most lists end within a few steps and are
  // then copied, with their initial points, into an arena block of their exact length); a list that outgrows the
  // stage moves into an arena block of the full capacity and grows there in place
  const uint32_t stage = cap < (uint32_t)EG3D_K3A_STAGE ? cap : (uint32_t)EG3D_K3A_STAGE;
  HPoint* scr = scratch + ((size_t)blockIdx.x * 64 + lane) * stage;
  uint32_t big_base = 0xffffffffu;  // != none: the list lives at arena[big_base ..]
  const uint32_t n_items = *n_items_p;  // the lists k3a_orient left to follow: (hypothesis, direction)
  bool have = false, exhausted = lane >= lanes_per_wave;
  uint32_t h = 0, dir = 0, n_init = 0, n_new = 0, init_off = 0, flags = 0;
  int st_start = 0;
  uint32_t dirs[3] = {0, 0, 0};
  int32_t ids[3] = {0, 0, 0};
  // views / polylines of the observations of the list's last point (entries 0-2) — a permutation of the
  // hypothesis' three (view, polyline) pairs
  int32_t cview[3] = {0, 0, 0};
  uint32_t cpl[3] = {0, 0, 0};
  K3A_T0();
  for (;;) {
    while (!have && !exhausted) {
      const uint32_t i = atomicAdd(queue, 1u);
      if (i >= n_items) {
        exhausted = true;
        break;
      }
      const uint32_t item = items[i];
      h = item >> 1;
      dir = item & 1u;
      const HypResult& r = res[h];
      const uint32_t stt = r.status;
      if (!(dir == 0 ? (stt & HYP_D1) : (stt & HYP_D2))) continue;
      n_init = dir == 0 ? r.n1 : r.n2;
      init_off = dir == 0 ? r.pts1_off : r.pts2_off;
      if (n_init == 0 || init_off == 0xffffffffu) continue;  // (arena overflow in the first phase)
      for (int k = 0; k < 3; k++) dirs[k] = dir == 0 ? r.dirs1[k] : r.dirs2[k];
      // (the points of the first phase list their observations in the order of the hypothesis' views: ascending)
      const HPoint last = arena[init_off + n_init - 1];
      for (int k = 0; k < 3; k++) {
        cview[k] = (int32_t)last.o[k].view;
        cpl[k] = last.o[k].pl;
        ids[k] = cview[k];
        L.load_polyline(s, k, cview[k], cpl[k]);
        PlPt p;
        p.seg = last.o[k].seg;
        p.x = last.o[k].x;
        p.y = last.o[k].y;
        L.set(k, p);
      }
      n_new = 0;
      flags = 0;
      st_start = 0;
      big_base = 0xffffffffu;
      have = true;
    }
    const unsigned long long working = __ballot(have);
    if (!working) break;
    const uint32_t K = k3a_slots_per_lane((uint32_t)__popcll(working));
    const uint32_t slot0 = (uint32_t)__popcll(working & lt) * K;
    sh.reqM[lane] = 0;
    __syncthreads();
    uint32_t n_req = 0, end_fl = 0;
    bool ended = false;
    if (have) {
      // requests of this iteration: the step at hand and, speculatively, the ones behind it. Never beyond the
      // probe of a list at capacity (one request past the last point that fits).
      const uint32_t total = n_init + n_new;
      const uint32_t room = total < cap ? cap - total : 0u;
      const uint32_t lim = K < room + 1u ? K : room + 1u;
      int32_t bview[3] = {cview[0], cview[1], cview[2]};
      uint32_t bpl[3] = {cpl[0], cpl[1], cpl[2]};
      L.set(3, L.get(0));
      L.set(4, L.get(1));
      L.set(5, L.get(2));
      int st0 = st_start;
      while (n_req < lim) {
        int st_used = 0;
        PlPt sel_pt[3];
        uint32_t fl = 0;
        if (!k3a_follow_request(s, L, bview, st0, dirs, ids, st_used, sel_pt, fl)) {
          ended = true;
          end_fl = fl;
          break;
        }
        const uint32_t slot = slot0 + n_req;
        int32_t nview[3];
        uint32_t npl[3];
        const int sel_idx[3] = {st_used, st_used == 0 ? 1 : 0, st_used == 2 ? 1 : 2};
        for (int k = 0; k < 3; k++) {
          nview[k] = k3a_pick(bview, sel_idx[k]);
          npl[k] = k3a_pick(bpl, sel_idx[k]);
          Obs o;
          o.view = (uint32_t)nview[k];
          o.pl = npl[k];
          o.seg = sel_pt[k].seg;
          o.x = sel_pt[k].x;
          o.y = sel_pt[k].y;
          sh.req[slot][k] = o;
        }
        sh.reqM[slot] = 1u | ((uint32_t)st_used << 8) | (fl << 16);
        n_req++;
        // the next request starts from this one's observations
        for (int k = 0; k < 3; k++) {
          bview[k] = nview[k];
          bpl[k] = npl[k];
          L.set(3 + k, sel_pt[k]);
        }
        st0 = 0;
      }
    }
    K3A_T(0);
    K3A_C(0, 1);
    K3A_C(2, __popcll(working));
    K3A_C(1, __popcll(__ballot(sh.reqM[lane] & 1u)));
    k3a_serve(sh, s, lane);
    K3A_T(1);
    if (have) {
      bool finish = false;
      uint32_t j = 0;
      for (; j < n_req; j++) {
        const uint32_t slot = slot0 + j;
        const uint32_t m = sh.reqM[slot], f = sh.resF[slot];
        flags |= (m >> 16) | (f >> 8);
        if (f & 1u) {
          if (n_init + n_new >= cap) {  // the probe of a full list: it would outgrow its capacity
            flags |= 4u;
            finish = true;
            break;
          }
          if (big_base != 0xffffffffu)
            arena[big_base + n_init + n_new] = k3a_point_of_slot(sh, slot);
          else
            scr[n_new] = k3a_point_of_slot(sh, slot);
          n_new++;
          if (big_base == 0xffffffffu && n_new == stage && n_init + n_new < cap) {  // outgrew the stage
            const uint32_t b = atomicAdd(&ctr->arena_used, cap);
            if (b + cap <= arena_cap) {
              for (uint32_t k = 0; k < n_init; k++) arena[b + k] = arena[init_off + k];
              for (uint32_t k = 0; k < n_new; k++) arena[b + n_init + k] = scr[k];
              big_base = b;
            } else {  // (the host redoes the stage with a larger arena: this list's result is not used)
              atomicOr(&ctr->flags, CTR_ARENA_OVERFLOW);
              n_new = 0;
              finish = true;
              break;
            }
          }
          for (int k = 0; k < 3; k++) {
            const Obs o = sh.req[slot][k];
            cview[k] = (int32_t)o.view;
            cpl[k] = o.pl;
            PlPt p;
            p.seg = o.seg;
            p.x = o.x;
            p.y = o.y;
            L.set(k, p);
          }
          st_start = 0;
        } else {
          st_start = (int)((m >> 8) & 3u) + 1;  // same step, next starting observation
          break;
        }
      }
      if (!finish && j == n_req && ended) {  // no observation left to start from: the list ends here
        flags |= end_fl;
        finish = true;
      }
      if (finish) {
        const uint32_t total = n_init + n_new;
        if (n_new && big_base != 0xffffffffu) {
          if (dir == 0) {
            res[h].pts1_off = big_base;
            res[h].n1 = total;
          } else {
            res[h].pts2_off = big_base;
            res[h].n2 = total;
          }
        } else if (n_new) {
          const uint32_t base = atomicAdd(&ctr->arena_used, total);
          if (base + total <= arena_cap) {
            for (uint32_t k = 0; k < n_init; k++) arena[base + k] = arena[init_off + k];
            for (uint32_t k = 0; k < n_new; k++) arena[base + n_init + k] = scr[k];
            if (dir == 0) {
              res[h].pts1_off = base;
              res[h].n1 = total;
            } else {
              res[h].pts2_off = base;
              res[h].n2 = total;
            }
          } else {
            atomicOr(&ctr->flags, CTR_ARENA_OVERFLOW);
          }
        }
        if (flags) {
          atomicOr(&ctr->flags, flags);
          atomicOr(&res[h].flags, flags);
        }
        have = false;
      }
    }
    K3A_T(2);
  }
  K3A_TEND(1);
}

}  // namespace eg3d
