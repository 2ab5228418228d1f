// eg3d_k3c_engine.h — the expand stage as a LANE-PER-CHAIN engine (gfx950 only; included by eg3d_kernels.hip).
//
// k3b_expand (rounds 1-4) gives a chain a whole wavefront: its Gauss-Newton solves and candidate searches spread over
// the 64 lanes, everything else — every side-walk step, every step of the chain following, every 2-view DLT, all the
// bookkeeping — is executed wave-uniformly, 64 lanes doing one chain's work. The counters of round 4 put that at 475 k
// vector instructions per C3' chain against ~90 k of useful work at full density
// (profiles/r05_experiments/stage0_chain_machine_counts.txt).
//
// Here ONE LANE owns a chain. The chain's program is the state machine of eg3d_chain_sm.h, whose only blocking points
// are batches of Gauss-Newton solves and of candidate searches. A wavefront is a small bulk-synchronous machine:
//
//   every iteration   (0) idle lanes take the next chain of the launch (longest first) and build its initial state,
//                     (1) every lane ADVANCES its machine to its next blocking point — lane-private walks, DLTs, commits,
//                     (2) lanes whose chain is finished pack it into the launch's staging area,
//                     (3) the wave DRAINS all pending solves: windows of EG3D_COOP_REQ requests, entries written by
//                         whichever lanes own them (each owner gets K = 32 / #owners slots per window), solved by the
//                         lane-group solver of eg3d_dev_coopgn.h — rows of different chains' requests side by side,
//                         ordered sums => the bits of the sequential solver — answers stored where the machines read them,
//                     (4) the wave DRAINS all pending candidate items as one flat sequence dealt to the 64 lanes.
//
// Results are those of the sequential statement whatever the interleaving: requests are pure functions of a chain's own
// state, and a machine only resumes when its whole batch is answered.
#pragma once
#include "eg3d_chain_sm.h"
#include "eg3d_dev_coopgn.h"

namespace eg3d {

#ifndef EG3D_K3C_WAVES
#define EG3D_K3C_WAVES 2 /* waves per SIMD the register allocation aims at (256 VGPRs) */
#endif

// LDS of a wave: the solver's block (CoopLds, 8 allocation units => four waves per SIMD fit) and, aliased onto parts of it
// that the engine does not use at the same time: the answer addresses of a window's entries in CoopLds::tmp_a (the
// N-view step's list of k3b_expand — the machine keeps its lists in the slice), the owner table of the candidate drain
// in the product columns (never live during a solve).
struct K3cOwners {
  unsigned char* slice[64];  // candidate drain: working slice, chain head, first point, view (| epi-only << 16) of each owner
  uint32_t excl[65];
  int32_t head[64], from[64], view[64];
};
struct K3cAnswers {
  float* resX[EG3D_COOP_REQ];  // where the answer of window entry j goes
  uint32_t* resOk[EG3D_COOP_REQ];
};
static_assert(sizeof(K3cOwners) <= sizeof(((CoopLds*)nullptr)->prod), "the owner table must fit the product columns");
static_assert(sizeof(K3cAnswers) <= sizeof(((CoopLds*)nullptr)->tmp_a), "the answer addresses must fit CoopLds::tmp_a");
struct K3cShared {
  CoopLds gn;
  __device__ __forceinline__ K3cOwners& own() { return *(K3cOwners*)&gn.prod[0][0]; }
  __device__ __forceinline__ K3cAnswers& ans() { return *(K3cAnswers*)&gn.tmp_a[0]; }
};

#if defined(EG3D_SECTION_TIMING) && !defined(EG3D_K3C_TIMING)
#define EG3D_K3C_TIMING 1
#endif
#ifdef EG3D_K3C_TIMING
// timing build (-DEG3D_K3C_TIMING: the engine's own clocks only, light; -DEG3D_SECTION_TIMING adds the solver's): shader clocks of a wave's phases summed over the launch — [0] fetch, [1] advance, [2] pack, [3] solve
// drain, [4] candidate drain; [8] iterations, [9] lanes with a chain (summed over iterations), [10] solves served,
// [11] candidate items, [12] solver windows, [13] lanes blocked on solves / [14] on candidates (summed over iterations)
__device__ unsigned long long g_k3c_dbg[32];
// ... and per block of sm_advance (index = SMS_* state): [3 s] wall clocks of the wave inside the block, [3 s + 1] the same
// times the lanes that were in it, [3 s + 2] entries
__device__ unsigned long long g_k3c_prof[96];
struct SmEnvProf : SmEnvStream {
  mutable unsigned long long t0 = 0;
  unsigned long long* acc = nullptr;  // the wave's 96 counters in LDS
  __device__ void prof_begin(uint32_t) const { t0 = __builtin_readcyclecounter(); }
  __device__ void prof_end(uint32_t id) const {
    const unsigned long long dt = __builtin_readcyclecounter() - t0;
    const unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63u) == __ffsll((long long)m) - 1) {
      acc[3 * id] += dt;
      acc[3 * id + 1] += dt * (unsigned long long)__popcll(m);
      acc[3 * id + 2] += 1ull;
    }
  }
};
#define K3C_ENV SmEnvProf
#define K3C_T0() unsigned long long ct_[6] = {0, 0, 0, 0, 0, 0}, cc_[7] = {0, 0, 0, 0, 0, 0, 0}, ct0_ = __builtin_readcyclecounter(), ct1_
#define K3C_T(i) (ct1_ = __builtin_readcyclecounter(), ct_[i] += ct1_ - ct0_, ct0_ = ct1_)
#define K3C_C(i, v) (cc_[i] += (v))
#define K3C_TEND()                                                                   \
  __syncthreads();                                                                   \
  for (uint32_t i_ = lane; i_ < 96; i_ += 64)                                        \
    if (prof_acc[i_]) atomicAdd(&g_k3c_prof[i_], prof_acc[i_]);                      \
  if (lane == 0) {                                                                   \
    for (int q_ = 0; q_ < 5; q_++) atomicAdd(&g_k3c_dbg[q_], ct_[q_]);               \
    for (int q_ = 0; q_ < 7; q_++) atomicAdd(&g_k3c_dbg[8 + q_], cc_[q_]);           \
  }
#else
#define K3C_ENV SmEnvStream
#define K3C_TEND() ((void)0)
#define K3C_T0() ((void)0)
#define K3C_T(i) ((void)0)
#define K3C_C(i, v) ((void)0)
#endif

template <bool LONG_GN>
__global__ void __launch_bounds__(64, EG3D_K3C_WAVES) k3c_engine_t(DevScene s, StageAView a, const TaskDesc* tasks,
                                                                  const ChainSeed* chains, uint32_t n_chains,
                                                                  const uint32_t* hyp_off, const HypResult* res,
                                                                  const HPoint* arena, const int32_t* map_view,
                                                                  const uint32_t* map_entry, const uint32_t* map_n,
                                                                  ChainLayout L, unsigned char* slices, StageBuf stage,
                                                                  ChainOut* outs, uint32_t* out_points, uint32_t* out_obs,
                                                                  Counters* ctr, const uint32_t* order, uint32_t* queue,
                                                                  uint32_t lanes_per_wave) {
  __shared__ K3cShared sh;
  const uint32_t lane = threadIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned char* const slice = slices + L.total * ((size_t)blockIdx.x * lanes_per_wave + (lane < lanes_per_wave ? lane : 0u));
  if (lane == 0) {
    sh.gn.cams_mid_range = s.cams_mid_range ? 1 : 0;
    sh.gn.long_refused = 0;
  }
  __syncthreads();
  SmChain q;
  q.k.wait = SM_DONE;
  q.k.gn_count = q.k.gn_issued = 0;
  q.k.cl_from = q.k.cl_to = 0;
  bool have = false, exhausted = lane >= lanes_per_wave;
  uint32_t jchain = 0, flags_acc = 0;
  unsigned long long bytes_acc = 0;
#ifdef EG3D_K3C_TIMING
  __shared__ unsigned long long prof_acc[96];
  for (uint32_t i_ = lane; i_ < 96; i_ += 64) prof_acc[i_] = 0;
  __syncthreads();
  K3C_ENV env;
  env.acc = prof_acc;
#else
  const K3C_ENV env;
#endif
  K3C_T0();
  for (;;) {
    // ---- (0) the next chain
    if (!have && !exhausted) {
      const uint32_t i = atomicAdd(queue, 1u);
      if (i >= n_chains) {
        exhausted = true;
      } else {
        jchain = order[i];
        const ChainSeed cs = chains[jchain];
        sm_begin(s, a, tasks[cs.task], cs, hyp_off[cs.task], res, arena, map_view, map_entry, map_n, L, slice,
                 (SmMbox*)(slice + L.off_mbox), q);
        have = true;
      }
    }
    if (!__ballot(have)) break;
    K3C_T(0);
    K3C_C(0, 1);
    K3C_C(1, __popcll(__ballot(have)));
    // ---- (1) advance
    if (have) sm_advance(env, s, a, q);
    K3C_T(1);
    K3C_C(5, __popcll(__ballot(have && q.k.wait == SM_WAIT_GN)));
    K3C_C(6, __popcll(__ballot(have && q.k.wait == SM_WAIT_CL)));
    // ---- (2) finished chains: pack (point headers + observations back to back), report
    if (have && q.k.wait == SM_DONE) {
      // (the point headers are requested eight at a time, the observations of a point four at a time: the lane packs alone)
      ChainOut co;
      const ChainPt* pts = q.c.pts + q.c.head;
      const uint32_t np = (uint32_t)q.c.len;
      uint32_t nobs_total = 0;
      for (uint32_t i0 = 0; i0 < np; i0 += 8) {
        uint32_t nn[8];
        for (uint32_t b = 0; b < 8; b++) nn[b] = i0 + b < np ? pts[i0 + b].nobs : 0u;
        for (uint32_t b = 0; b < 8; b++) nobs_total += nn[b];
      }
      co.n_points = np;
      co.n_obs = nobs_total;
      co.flags = q.c.flags;
      co.head = (uint32_t)q.c.head;
      co.bytes = q.c.bytes;
      for (int k = 0; k < 16; k++) co.tsec[k] = 0;
      const unsigned long long pb = atomicAdd(&stage.used[0], (unsigned long long)co.n_points);
      const unsigned long long ob = atomicAdd(&stage.used[1], (unsigned long long)co.n_obs);
      co.spt = pb;
      co.sobs = ob;
      if (pb + co.n_points <= stage.cap_pts && ob + co.n_obs <= stage.cap_obs) {
        StagePt* spt = stage.pts + pb;
        Obs* sob = stage.obs + ob;
        for (uint32_t i0 = 0; i0 < np; i0 += 4) {
          ChainPt hp[4];
          for (uint32_t b = 0; b < 4; b++) hp[b] = pts[i0 + b < np ? i0 + b : i0];
          for (uint32_t b = 0; b < 4 && i0 + b < np; b++) {
            StagePt sp;
            sp.X[0] = hp[b].X[0];
            sp.X[1] = hp[b].X[1];
            sp.X[2] = hp[b].X[2];
            sp.nobs = hp[b].nobs;
            spt[i0 + b] = sp;
            sm_copy_obs(sob, q.c.pool + hp[b].off, hp[b].nobs);
            sob += hp[b].nobs;
          }
        }
      }
      outs[jchain] = co;
      out_points[jchain] = co.n_points;
      out_obs[jchain] = co.n_obs;
      flags_acc |= co.flags;
      bytes_acc += co.bytes;
      have = false;
    }
    K3C_T(2);
    // ---- (3) drain the solves
    for (;;) {
      const bool has = have && q.k.wait == SM_WAIT_GN && q.k.gn_issued < q.k.gn_count;
      const unsigned long long mask = __ballot(has);
      if (!mask) break;
      const uint32_t nl = (uint32_t)__popcll(mask);
      uint32_t K = 1;
      while (K * 2 * nl <= (uint32_t)EG3D_COOP_REQ) K *= 2;
      const uint32_t r = (uint32_t)__popcll(mask & lt);
      const bool taking = has && r * K < (uint32_t)EG3D_COOP_REQ;
      if (lane < EG3D_COOP_REQ) {
        sh.gn.n16[lane] = 0;
        sh.gn.res_ok[lane] = 0;
      }
      __syncthreads();
      if (taking) {
        uint32_t filled = 0;
        while (filled < K && q.k.gn_issued < q.k.gn_count) {
          SmGnReq rq;
          if (sm_gn_request(q, q.k.gn_issued, rq)) {
            const uint32_t slot = r * K + filled;
            sh.gn.gbase[slot] = rq.base;
            sh.gn.n16[slot] = (uint16_t)((uint32_t)(rq.nblock + (rq.has_extra ? 1 : 0)) | (rq.has_extra ? 0x8000u : 0u));
            sh.gn.ex_view[slot] = rq.ex_view;
            sh.gn.ex_x[slot] = rq.ex_x;
            sh.gn.ex_y[slot] = rq.ex_y;
            sh.gn.x0[slot][0] = rq.X0[0];
            sh.gn.x0[slot][1] = rq.X0[1];
            sh.gn.x0[slot][2] = rq.X0[2];
            sh.ans().resX[slot] = rq.resX;
            sh.ans().resOk[slot] = rq.resOk;
            filled++;
          }
          q.k.gn_issued++;
        }
      }
      __syncthreads();
      {
        const int n_req = lane < EG3D_COOP_REQ ? (int)(sh.gn.n16[lane] & 0x7fff) : 0;
        K3C_C(2, __popcll(__ballot(n_req != 0)));
        K3C_C(4, 1);
        float Xr[3];
        coop_gn_run<0, LONG_GN, true>(s.cam_P, sh.gn, n_req != 0, n_req, Xr);
        // answers: the table keeps them until the next window
        if (n_req != 0) {
          const uint32_t ok = sh.gn.res_ok[lane];
          *sh.ans().resOk[lane] = ok;
          if (ok) {
            float* X = sh.ans().resX[lane];
            X[0] = sh.gn.x0[lane][0];
            X[1] = sh.gn.x0[lane][1];
            X[2] = sh.gn.x0[lane][2];
          }
        }
      }
      __syncthreads();  // (answers stored: s_waitcnt vmcnt(0) + barrier, as between a solver batch and its readers in k3b_expand)
    }
    if (have && q.k.wait == SM_WAIT_GN) q.k.wait = SM_RUN;
    K3C_T(3);
    // ---- (4) drain the candidate items: all of them as one flat sequence
    {
      const int cnt = (have && q.k.wait == SM_WAIT_CL) ? q.k.cl_to - q.k.cl_from : 0;
      if (__ballot(cnt > 0)) {
        const int incl = wave_incl_scan(cnt);
        const int total = lane_bcast(incl, 63);
        K3cOwners& ow = sh.own();
        ow.excl[lane] = (uint32_t)(incl - cnt);
        if (lane == 63) ow.excl[64] = (uint32_t)total;
        ow.slice[lane] = slice;
        ow.head[lane] = q.c.head;
        ow.from[lane] = q.k.cl_from;
        ow.view[lane] = q.k.v | (int32_t)(q.k.cl_epi_only << 16);
        __syncthreads();
        K3C_C(3, total);
        for (int f0 = 0; f0 < total; f0 += 64) {
          const int f = f0 + (int)lane;
          if (f < total) {
            uint32_t lo = 0;  // the owner whose range holds f: largest o with excl[o] <= f (owners without items skipped)
#pragma unroll
            for (uint32_t step = 32; step; step >>= 1)
              if (ow.excl[lo + step] <= (uint32_t)f) lo += step;
            unsigned char* const sl = ow.slice[lo];
            const int vw = ow.view[lo];
            const int idx = ow.from[lo] + (f - (int)ow.excl[lo]);
            if (vw >> 16)
              sm_epiline_item(s, (const ChainPt*)(sl + L.off_pts), (const Obs*)(sl + L.off_pool), (ViewCand*)(sl + L.off_cand),
                              ow.head[lo], vw & 0xffff, idx);
            else
              sm_closest_item(s, (const ChainPt*)(sl + L.off_pts), (const Obs*)(sl + L.off_pool), (ViewCand*)(sl + L.off_cand),
                              ow.head[lo], vw & 0xffff, idx);
          }
        }
        __syncthreads();
      }
      if (have && q.k.wait == SM_WAIT_CL) q.k.wait = SM_RUN;
    }
    K3C_T(4);
  }
  // ---- the wave's flags and byte count: one atomic each
  {
    uint32_t fl = 0;
#pragma unroll
    for (uint32_t b = 1; b <= 16u; b <<= 1)
      if (__ballot((flags_acc & b) != 0)) fl |= b;
    unsigned long long bsum = bytes_acc;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) bsum += (unsigned long long)__shfl_xor((long long)bsum, d);
    if (lane == 0) {
      if (fl) atomicOr(&ctr->flags, fl);
      if (!LONG_GN && sh.gn.long_refused) atomicOr(&ctr->flags, CTR_LONG_REFUSED);
      if (bsum) atomicAdd(&ctr->bytes, bsum);
    }
  }
  K3C_TEND();
}

}  // namespace eg3d
