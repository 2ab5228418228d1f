// TEST-ONLY host simulation of the stage-B kernels (not part of the product, never linked
// into libeg3d.so). It compiles the per-lane device bodies of edgegraph3d_amd/csrc/*.h with
// g++ and runs them serially, one "lane" at a time, in the same order of phases as the GPU
// pipeline (task_setup -> hypotheses -> select -> expand -> emit). Purpose: debug the kernel
// logic against the CPU oracle on machines without a GPU. Stage A (K1/K2) is wave-cooperative
// HIP code and cannot run here; the harness takes stage-A results as input.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/eg3d.h"
#include "../../include/eg3d_host.h"
static unsigned long long g_stat[8], g_hist[2][16], g_hsum[2][16];
#define EG3D_STAT(i) (g_stat[i]++)
#include "eg3d_dev_pipeline.h"
#include "eg3d_chain_sm.h"

using namespace eg3d;

struct HostScene {
  DevScene ds;
  std::vector<uint32_t> g30_off, g30_ids, g4_off, g4_ids;
};

static int build_dev_scene(const eg3d_scene* sc, HostScene& hs) {
  DevScene& d = hs.ds;
  d.n_views = sc->n_views;
  d.width = sc->width;
  d.height = sc->height;
  d.cam_P = sc->cam_P;
  d.F = sc->F;
  d.F_valid = sc->F_valid;
  d.view_pl_off = sc->view_pl_off;
  d.pl_vtx_off = sc->pl_vtx_off;
  d.vtx = reinterpret_cast<const f2*>(sc->vtx_xy);
  d.pl_start = sc->pl_start;
  d.pl_end = sc->pl_end;
  for (int which = 0; which < 2; which++) {
    std::vector<uint32_t>& off = which == 0 ? hs.g30_off : hs.g4_off;
    std::vector<uint32_t>& ids = which == 0 ? hs.g30_ids : hs.g4_ids;
    off.assign(1, 0);
    uint32_t w = 0, h = 0;
    for (int v = 0; v < sc->n_views; v++) {
      uint32_t *o, *i, dropped;
      if (eg3d_host_build_grid(sc, v, which == 0 ? 30.0f : 4.0f, &w, &h, &o, &i, &dropped) != 0) return -1;
      uint32_t base = (uint32_t)ids.size();
      for (uint32_t c = 0; c < w * h; c++) off.push_back(base + o[c + 1]);
      ids.insert(ids.end(), i, i + o[w * h]);
      free(o);
      free(i);
    }
    if (which == 0) {
      d.g30_w = (int)w;
      d.g30_h = (int)h;
    } else {
      d.g4_w = (int)w;
      d.g4_h = (int)h;
    }
  }
  d.g30_off = hs.g30_off.data();
  d.g30_ids = hs.g30_ids.data();
  d.g4_off = hs.g4_off.data();
  d.g4_ids = hs.g4_ids.data();
  return 0;
}

// The expand stage as the chain STATE MACHINE of eg3d_chain_sm.h (what the engine kernel k3c runs, one lane per
// chain) with a sequential server: requests are answered one after the other by the plain solver / candidate code.
static unsigned long long g_sm[16], g_smreq[8], g_smrows[8];  // g_smreq / g_smrows: solves / rows by batch kind  // advances, GN batches, GN requests, rows, closest batches, closest items, by batch kind [8..]
template <class Env>
static void run_chain_machine(const DevScene& ds, const StageAView& a, const TaskDesc& d, const ChainSeed& cs, uint32_t hyp_base,
                              const HypResult* res, const HPoint* arena, const int32_t* map_view, const uint32_t* map_entry,
                              const uint32_t* map_n, const ChainLayout& L, unsigned char* slice, ChainOut& out) {
  SmChain q;
  sm_begin(ds, a, d, cs, hyp_base, res, arena, map_view, map_entry, map_n, L, slice, (SmMbox*)(slice + L.off_mbox), q);
  const Env env;
  for (;;) {
    sm_advance(env, ds, a, q);
    g_sm[0]++;
    if (q.k.wait == SM_DONE) break;
    if (q.k.wait == SM_WAIT_GN) {
      g_sm[1]++;
      g_sm[8 + (q.k.gn_kind & 7)]++;
      for (int j = q.k.gn_issued; j < q.k.gn_count; j++) {
        SmGnReq r;
        if (!sm_gn_request(q, j, r)) continue;
        g_sm[2]++;
        g_sm[3] += (unsigned long long)(r.nblock + (r.has_extra ? 1 : 0));
        g_smreq[q.k.gn_kind & 7]++;
        g_smrows[q.k.gn_kind & 7] += (unsigned long long)(r.nblock + (r.has_extra ? 1 : 0));
        Obs ex;
        ex.view = (uint32_t)r.ex_view;
        ex.pl = 0;
        ex.seg = 0;
        ex.x = r.ex_x;
        ex.y = r.ex_y;
        ArrayCursor cur;
        cur.a = r.base;
        cur.n = r.nblock;
        cur.extra = r.has_extra ? &ex : nullptr;
        cur.i = 0;
        const double X0[3] = {(double)r.X0[0], (double)r.X0[1], (double)r.X0[2]};
        float X[3] = {0.f, 0.f, 0.f};
        const bool ok = gauss_newton_f64(ds.cam_P, cur, X0, X);
        *r.resOk = ok ? 1u : 0u;
        if (ok) {
          r.resX[0] = X[0];
          r.resX[1] = X[1];
          r.resX[2] = X[2];
        }
      }
      q.k.gn_issued = q.k.gn_count;
      q.k.wait = SM_RUN;
    } else if (q.k.wait == SM_WAIT_CL) {
      g_sm[4]++;
      for (int i = q.k.cl_from; i < q.k.cl_to; i++) {
        if (q.k.cl_epi_only)
          sm_epiline_item(ds, q.c.pts, q.c.pool, q.c.cand, q.c.head, q.k.v, i);
        else
          sm_closest_item(ds, q.c, q.k.v, i);
        g_sm[5]++;
      }
      q.k.wait = SM_RUN;
    }
  }
  sm_finish(q, out);
}

extern "C" int hostsim_match(const eg3d_scene* sc, const eg3d_seeds* seeds, uint32_t b, uint32_t e,
                             const eg3d_candidates* ca, uint32_t hyp_cap, uint32_t chain_cap, uint32_t pool_cap,
                             int slot_step, eg3d_edgepoints* out) {
  memset(out, 0, sizeof(*out));
  HostScene hs;
  if (build_dev_scene(sc, hs) != 0) return -1;
  const DevScene& ds = hs.ds;
  const uint32_t sv_base = seeds->trk_off[b];
  const uint32_t n_sv = seeds->trk_off[e] - sv_base;
  if (ca->n_sv != n_sv) return -2;
  // sv -> (seed, entry)
  std::vector<uint32_t> sv_seed(n_sv), sv_entry(n_sv);
  for (uint32_t s = b; s < e; s++)
    for (uint32_t i = seeds->trk_off[s]; i < seeds->trk_off[s + 1]; i++) {
      sv_seed[i - sv_base] = s;
      sv_entry[i - sv_base] = i - seeds->trk_off[s];
    }
  const uint32_t nt = ca->n_tasks;
  std::vector<uint32_t> task_seed(nt), task_entry(nt), list_ptr, list_cnt;
  std::vector<Obs> hits;
  const uint32_t n_lists = nt ? ca->task_list_off[nt] : 0;
  list_ptr.resize(n_lists);
  list_cnt.resize(n_lists);
  for (uint32_t t = 0; t < nt; t++) {
    task_seed[t] = sv_seed[ca->task_sv[t]];
    task_entry[t] = sv_entry[ca->task_sv[t]];
    const uint32_t l0 = ca->task_list_off[t], l1 = ca->task_list_off[t + 1];
    for (uint32_t l = l0; l < l1; l++) {
      const int32_t view = seeds->trk_view[seeds->trk_off[task_seed[t]] + (l - l0)];
      list_ptr[l] = (uint32_t)hits.size();
      list_cnt[l] = ca->list_off[l + 1] - ca->list_off[l];
      for (uint32_t h = ca->list_off[l]; h < ca->list_off[l + 1]; h++) {
        Obs o;
        o.view = view;
        o.pl = ca->hit_pl[h];
        o.seg = ca->hit_seg[h];
        o.x = ca->hit_xy[2 * h];
        o.y = ca->hit_xy[2 * h + 1];
        hits.push_back(o);
      }
    }
  }
  StageAView a;
  a.trk_off = seeds->trk_off;
  a.trk_view = seeds->trk_view;
  a.trk_xy = seeds->trk_xy;
  a.seed_begin = b;
  a.sv_base = sv_base;
  a.n_tasks = nt;
  a.task_seed = task_seed.data();
  a.task_entry = task_entry.data();
  a.task_hit = ca->task_hit;
  a.task_list_off = ca->task_list_off;
  a.list_ptr = list_ptr.data();
  a.list_cnt = list_cnt.data();
  a.hits = hits.data();
  // seed view maps
  std::vector<int32_t> map_view(n_sv ? n_sv : 1);
  std::vector<uint32_t> map_entry(n_sv ? n_sv : 1), map_n(e - b ? e - b : 1);
  for (uint32_t s = b; s < e; s++) {
    uint32_t base = seeds->trk_off[s] - sv_base;
    map_n[s - b] = build_seed_view_map(seeds->trk_view + seeds->trk_off[s], seeds->trk_off[s + 1] - seeds->trk_off[s],
                                       map_view.data() + base, map_entry.data() + base);
  }
  // task setup + hypothesis offsets
  std::vector<TaskDesc> tasks(nt ? nt : 1);
  std::vector<uint32_t> hyp_off(nt + 1, 0);
  for (uint32_t t = 0; t < nt; t++) {
    task_setup(a, t, map_view.data(), map_entry.data(), map_n.data(), tasks[t]);
    hyp_off[t + 1] = hyp_off[t] + tasks[t].n_hyp;
  }
  const uint32_t n_hyp = hyp_off[nt];
  for (int i = 0; i < 8; i++) g_stat[i] = 0;
  for (int i = 0; i < 16; i++) g_sm[i] = 0;
  for (int i = 0; i < 8; i++) g_smreq[i] = g_smrows[i] = 0;
  // K3a
  std::vector<HypResult> res(n_hyp ? n_hyp : 1);
  std::vector<HPoint> arena;
  std::vector<HPoint> s1(hyp_cap), s2(hyp_cap);
  uint32_t flags = 0;
  for (uint32_t t = 0; t < nt; t++)
    for (uint32_t h = hyp_off[t]; h < hyp_off[t + 1]; h++) {
      Obs c[3];
      hypothesis_hits(a, tasks[t], t, h - hyp_off[t], c);
      const unsigned long long tri_before = g_stat[1], ntri_before = g_stat[3];
      evaluate_hypothesis(HTeamSeq(), ds, c, s1.data(), s2.data(), hyp_cap, res[h]);
      {  // sequential triangulations of this hypothesis: orientation phase / following (histograms by powers of two)
        unsigned long long a1 = g_stat[1] - tri_before, a2 = g_stat[3] - ntri_before;
        int b1 = 0, b2 = 0;
        while ((1ull << b1) <= a1 && b1 < 15) b1++;
        while ((1ull << b2) <= a2 && b2 < 15) b2++;
        g_hist[0][b1]++;
        g_hist[1][b2]++;
        g_hsum[0][b1] += a1;
        g_hsum[1][b2] += a2;
      }
      flags |= res[h].flags;
      if (res[h].status & HYP_COMPAT) {
        res[h].pts1_off = (uint32_t)arena.size();
        arena.insert(arena.end(), s1.begin(), s1.begin() + res[h].n1);
      }
      if (res[h].status & HYP_D2) {
        res[h].pts2_off = (uint32_t)arena.size();
        arena.insert(arena.end(), s2.begin(), s2.begin() + res[h].n2);
      }
    }
  if (getenv("HOSTSIM_HYP_STATS")) {  // outcome histogram of the hypothesis stage (a tool for sizing K3a's phases)
    uint64_t n_tri = 0, n_d1 = 0, n_d2 = 0, n_compat = 0, sum_n1 = 0, sum_n2 = 0, h1[8] = {0}, follow_items = 0;
    for (uint32_t h = 0; h < n_hyp; h++) {
      const uint32_t st = res[h].status;
      n_tri += !!(st & HYP_TRI);
      n_d1 += !!(st & HYP_D1);
      n_d2 += !!(st & HYP_D2);
      n_compat += !!(st & HYP_COMPAT);
      if (st & HYP_D1) {
        sum_n1 += res[h].n1;
        h1[res[h].n1 < 7 ? res[h].n1 : 7]++;
        follow_items++;
      }
      if (st & HYP_D2) {
        sum_n2 += res[h].n2;
        follow_items++;
      }
    }
    fprintf(stderr,
            "hostsim hyp stats: tasks %u hyp %u tri %llu d1 %llu d2 %llu compat %llu sum_n1 %llu sum_n2 %llu follow_items %llu "
            "n1 histogram",
            nt, n_hyp, (unsigned long long)n_tri, (unsigned long long)n_d1, (unsigned long long)n_d2,
            (unsigned long long)n_compat, (unsigned long long)sum_n1, (unsigned long long)sum_n2,
            (unsigned long long)follow_items);
    for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", (unsigned long long)h1[i]);
    fprintf(stderr, "; calls: step3 %llu (its triangulations %llu) stepn3 starts %llu (its triangulations %llu)", g_stat[0],
            g_stat[1], g_stat[2], g_stat[3]);
    fprintf(stderr, "; line walks %llu, segments scanned beyond the first %llu (%.1f per walk)", g_stat[5], g_stat[6],
            (double)g_stat[6] / (double)(g_stat[5] ? g_stat[5] : 1));
    for (int w = 0; w < 2; w++) {
      fprintf(stderr, "\n  %s: hypotheses (sum of triangulations) with [0], [1], [2-3], [4-7] ... of them:", w ? "following" : "orientation");
      for (int i = 0; i < 12; i++) fprintf(stderr, " %llu (%llu)", g_hist[w][i], g_hsum[w][i]);
    }
    fprintf(stderr, "\n");
  }
  // K3s
  std::vector<ChainSeed> chains;
  for (uint32_t t = 0; t < nt; t++) {
    ChainSeed cs;
    cs.task = t;
    if (hyp_off[t + 1] > hyp_off[t] && select_task(res.data(), hyp_off[t], hyp_off[t + 1], cs)) chains.push_back(cs);
  }
  // K3b
  ChainLayout L = chain_layout(chain_cap, pool_cap, (uint32_t)ds.n_views);
  std::vector<unsigned char> scratch(L.total * (chains.size() ? chains.size() : 1));
  std::vector<ChainOut> couts(chains.size() ? chains.size() : 1);
  uint64_t np = 0, no = 0;
  const unsigned long long sm_walks0 = g_stat[5], sm_segs0 = g_stat[6];
  for (size_t j = 0; j < chains.size(); j++) {
    const ChainSeed& cs = chains[j];
    if (slot_step == 3)  // the machine with the streaming forms of its lane-private loops (what the engine kernel runs)
      run_chain_machine<SmEnvStream>(ds, a, tasks[cs.task], cs, hyp_off[cs.task], res.data(), arena.data(), map_view.data(),
                                     map_entry.data(), map_n.data(), L, scratch.data() + L.total * j, couts[j]);
    else if (slot_step == 2)
      run_chain_machine<SmEnvSeq>(ds, a, tasks[cs.task], cs, hyp_off[cs.task], res.data(), arena.data(), map_view.data(),
                        map_entry.data(), map_n.data(), L, scratch.data() + L.total * j, couts[j]);
    else if (slot_step)
      expand_chain(TeamSeqSlots(), ds, a, tasks[cs.task], cs, hyp_off[cs.task], res.data(), arena.data(),
                   map_view.data(), map_entry.data(), map_n.data(), L, scratch.data() + L.total * j, couts[j]);
    else
      expand_chain(TeamSeq(), ds, a, tasks[cs.task], cs, hyp_off[cs.task], res.data(), arena.data(), map_view.data(),
                   map_entry.data(), map_n.data(), L, scratch.data() + L.total * j, couts[j]);
    flags |= couts[j].flags;
    np += couts[j].n_points;
    no += couts[j].n_obs;
  }
  if (slot_step == 2 && getenv("HOSTSIM_SM_STATS")) {  // stage 0 of the engine: what the chains ask for, per chain
    const double nc = chains.size() ? (double)chains.size() : 1.0;
    fprintf(stderr,
            "hostsim chain machine: %zu chains, %llu points; per chain: %.1f blocking points (GN batches %.1f, CLOSEST batches %.1f), "
            "%.1f solves of %.2f rows, %.1f closest items; GN batches by kind epc %.2f presolve %.2f central %.2f sides %.2f list %.2f "
            "listadd %.2f; line walks %.1f (+%.2f segments beyond the first each)\n",
            chains.size(), (unsigned long long)np, g_sm[0] / nc, g_sm[1] / nc, g_sm[4] / nc, g_sm[2] / nc,
            g_sm[2] ? (double)g_sm[3] / (double)g_sm[2] : 0.0, g_sm[5] / nc, g_sm[8 + SMB_EPC] / nc, g_sm[8 + SMB_PRESOLVE] / nc,
            g_sm[8 + SMB_CENTRAL] / nc, g_sm[8 + SMB_SIDES] / nc, (g_sm[8 + SMB_LIST_A] + g_sm[8 + SMB_LIST_B]) / nc, g_sm[8 + SMB_LISTADD] / nc,
            (g_stat[5] - sm_walks0) / nc, (g_stat[5] - sm_walks0) ? (double)(g_stat[6] - sm_segs0) / (double)(g_stat[5] - sm_walks0) : 0.0);
  }
  if (slot_step >= 2 && getenv("HOSTSIM_SM_STATS")) {
    const double nc = chains.size() ? (double)chains.size() : 1.0;
    static const char* kn[8] = {"", "epc", "presolve", "central", "sides", "list", "listadd", "list3"};
    fprintf(stderr, "  solves per chain by kind (rows each):");
    for (int k = 1; k < 8; k++)
      if (g_smreq[k]) fprintf(stderr, " %s %.1f (%.1f)", kn[k], g_smreq[k] / nc, (double)g_smrows[k] / (double)g_smreq[k]);
    fprintf(stderr, "\n");
  }
  // K4
  out->n_points = np;
  out->n_obs = no;
  out->X = (float*)malloc(sizeof(float) * 3 * (np ? np : 1));
  out->obs_off = (uint64_t*)malloc(sizeof(uint64_t) * (np + 1));
  out->obs_view = (int32_t*)malloc(sizeof(int32_t) * (no ? no : 1));
  out->obs_pl = (uint32_t*)malloc(sizeof(uint32_t) * (no ? no : 1));
  out->obs_seg = (uint32_t*)malloc(sizeof(uint32_t) * (no ? no : 1));
  out->obs_xy = (float*)malloc(sizeof(float) * 2 * (no ? no : 1));
  out->key = (uint32_t*)malloc(sizeof(uint32_t) * 4 * (np ? np : 1));
  uint64_t pb = 0, ob = 0;
  for (size_t j = 0; j < chains.size(); j++) {
    emit_chain(L, scratch.data() + L.total * j, couts[j], tasks[chains[j].task], pb, ob, out->X, out->obs_off,
               out->obs_view, out->obs_pl, out->obs_seg, out->obs_xy, out->key);
    pb += couts[j].n_points;
    ob += couts[j].n_obs;
  }
  out->obs_off[np] = (uint64_t)no;
  out->n_tasks = nt;
  out->n_hypotheses = n_hyp;
  out->n_chains = chains.size();
  out->flags = flags;
  return 0;
}

extern "C" void hostsim_free_edgepoints(eg3d_edgepoints* e) {
  free(e->X);
  free(e->obs_off);
  free(e->obs_view);
  free(e->obs_pl);
  free(e->obs_seg);
  free(e->obs_xy);
  free(e->key);
  memset(e, 0, sizeof(*e));
}

// ---- primitive probes so the device bodies can be KAT-tested on CPU too ----
// polyline_closest_pruned (block bounding-box pre-test) against the plain scan polyline_closest_range on one
// polyline and many query points: returns the number of queries whose (distance bits, segment, x, y) differ.
extern "C" int hostsim_closest_pruned_mismatches(const float* vtx, int n, const float* pts, int n_pts, uint32_t s0,
                                                 uint32_t s1) {
  std::vector<float> bb;
  const uint32_t nseg = n > 1 ? (uint32_t)n - 1u : 0u;
  for (uint32_t b0 = 0; b0 < nseg; b0 += EG3D_BB_SEGS) {
    const uint32_t b1 = b0 + EG3D_BB_SEGS < nseg ? b0 + EG3D_BB_SEGS : nseg;
    float x0 = vtx[2 * b0], y0 = vtx[2 * b0 + 1], x1 = x0, y1 = y0;
    for (uint32_t i = b0 + 1; i <= b1; i++) {
      x0 = vtx[2 * i] < x0 ? vtx[2 * i] : x0;
      y0 = vtx[2 * i + 1] < y0 ? vtx[2 * i + 1] : y0;
      x1 = vtx[2 * i] > x1 ? vtx[2 * i] : x1;
      y1 = vtx[2 * i + 1] > y1 ? vtx[2 * i + 1] : y1;
    }
    bb.insert(bb.end(), {x0, y0, x1, y1});
  }
  PlRef plain, boxed;
  plain.v = boxed.v = (const f2*)vtx;
  plain.n = boxed.n = (uint32_t)n;
  plain.start = boxed.start = 0;
  plain.end = boxed.end = 1;
  boxed.bb = bb.empty() ? nullptr : bb.data();
  int bad = 0;
  for (int k = 0; k < n_pts; k++) {
    PlPt a, b;
    const float da = polyline_closest_range(plain, pts[2 * k], pts[2 * k + 1], s0, s1, a);
    const float db = polyline_closest_pruned(boxed, pts[2 * k], pts[2 * k + 1], s0, s1, b);
    if (memcmp(&da, &db, 4) != 0 || a.seg != b.seg || memcmp(&a.x, &b.x, 4) != 0 || memcmp(&a.y, &b.y, 4) != 0) bad++;
  }
  return bad;
}

extern "C" float hostsim_dist2(float ax, float ay, float bx, float by) { return dist2(ax, ay, bx, by); }
extern "C" float hostsim_seg_closest(float px, float py, float vx, float vy, float wx, float wy, float* q) {
  return seg_closest(px, py, vx, vy, wx, wy, q[0], q[1]);
}
extern "C" int hostsim_walk_by_distance(const float* vtx, int n, uint32_t start, uint32_t end, uint32_t seg, float x,
                                        float y, uint32_t direction, float distance, uint32_t* oseg, float* oxy) {
  PlRef pl;
  pl.v = reinterpret_cast<const f2*>(vtx);
  pl.n = (uint32_t)n;
  pl.start = start;
  pl.end = end;
  PlPt p, o;
  p.seg = seg;
  p.x = x;
  p.y = y;
  uint32_t w = walk_by_distance(pl, p, direction, distance, o);
  *oseg = o.seg;
  oxy[0] = o.x;
  oxy[1] = o.y;
  return (int)w;
}
extern "C" int hostsim_walk_by_line(const float* vtx, int n, uint32_t start, uint32_t end, uint32_t seg, float x,
                                    float y, uint32_t direction, const float* line, int bounded, float mind,
                                    float maxd, uint32_t* oseg, float* oxy) {
  PlRef pl;
  pl.v = reinterpret_cast<const f2*>(vtx);
  pl.n = (uint32_t)n;
  pl.start = start;
  pl.end = end;
  PlPt p, o;
  p.seg = seg;
  p.x = x;
  p.y = y;
  o = p;
  uint32_t w = walk_by_line(pl, p, direction, line[0], line[1], line[2], bounded != 0, mind, maxd, o);
  *oseg = o.seg;
  oxy[0] = o.x;
  oxy[1] = o.y;
  return (int)w;
}
// the walks with their loads in flight (walk_by_*_pf) against the plain walks: every start segment of the polyline,
// both directions and a direction that is neither end; n_q queries (a point near the polyline + a line / a
// distance each). Returns the number of mismatches (status, segment, coordinate bits).
extern "C" int hostsim_walk_pf_mismatches(const float* vtx, int n, uint32_t start, uint32_t end, const float* q, int n_q) {
  PlRef pl;
  pl.v = reinterpret_cast<const f2*>(vtx);
  pl.n = (uint32_t)n;
  pl.start = start;
  pl.end = end;
  int bad = 0;
  const uint32_t dirs[3] = {start, end, 0xfffffff0u};
  for (int i = 0; i < n_q; i++) {
    const float* qi = q + 8 * i;  // seg (as float), x, y, la, lb, lc, distance, bounded
    PlPt p;
    p.seg = (uint32_t)qi[0];
    p.x = qi[1];
    p.y = qi[2];
    for (int d = 0; d < 3; d++) {
      PlPt a = p, b = p;
      const uint32_t wa = walk_by_line(pl, p, dirs[d], qi[3], qi[4], qi[5], qi[7] != 0.0f, 5.0f, 20.0f, a);
      const uint32_t wb = walk_by_line_pf(pl, p, dirs[d], qi[3], qi[4], qi[5], qi[7] != 0.0f, 5.0f, 20.0f, b);
      if (wa != wb || ((wa & (WALK_FOUND | WALK_BOUND)) && (a.seg != b.seg || memcmp(&a.x, &b.x, 4) || memcmp(&a.y, &b.y, 4)))) bad++;
      PlPt c = p, e = p;
      const uint32_t wc = walk_by_distance(pl, p, dirs[d], qi[6], c);
      const uint32_t we = walk_by_distance_pf(pl, p, dirs[d], qi[6], e);
      if (wc != we || c.seg != e.seg || memcmp(&c.x, &e.x, 4) || memcmp(&c.y, &e.y, 4)) bad++;
    }
  }
  return bad;
}
extern "C" int hostsim_triangulate(const float* cam_P, const int32_t* views, const float* xy, int n, float* X) {
  std::vector<Obs> a(n);
  for (int i = 0; i < n; i++) {
    a[i].view = views[i];
    a[i].pl = 0;
    a[i].seg = 0;
    a[i].x = xy[2 * i];
    a[i].y = xy[2 * i + 1];
  }
  uint32_t flags = 0;
  return triangulate_array(cam_P, a.data(), n, X, flags) ? 1 : 0;
}
// diagnostics (tools/c5_iterations.py): residual passes each point of the config-5 filter runs (1..30; the device code
// of k5_gn_filter compiled for the host, so the counts are the kernel's)
extern "C" int hostsim_gn_filter_iters(const float* cam_P, const float* X, const uint32_t* obs_off, const int32_t* obs_view,
                                       const float* obs_xy, uint64_t n, float gn_max_mse, int legacy_abs, uint8_t* iters) {
#pragma omp parallel for schedule(static, 4096)
  for (int64_t i = 0; i < (int64_t)n; i++) {
    uint32_t a = obs_off[i], b = obs_off[i + 1];
    float o[3];
    int it = 0;
    (void)gauss_newton_f32(cam_P, obs_view + a, obs_xy + 2 * a, (int)(b - a), X + 3 * i, gn_max_mse, legacy_abs != 0, o, &it);
    iters[i] = (uint8_t)it;
  }
  return 0;
}

extern "C" int hostsim_gn_filter(const float* cam_P, const float* X, const uint32_t* obs_off, const int32_t* obs_view,
                                 const float* obs_xy, uint64_t n, float gn_max_mse, int legacy_abs, float* Xo,
                                 uint8_t* inl) {
  for (uint64_t i = 0; i < n; i++) {
    uint32_t a = obs_off[i], b = obs_off[i + 1];
    float o[3];
    bool ok = gauss_newton_f32(cam_P, obs_view + a, obs_xy + 2 * a, (int)(b - a), X + 3 * i, gn_max_mse,
                               legacy_abs != 0, o);
    inl[i] = ok;
    for (int k = 0; k < 3; k++) Xo[3 * i + k] = ok ? o[k] : X[3 * i + k];
  }
  return 0;
}
