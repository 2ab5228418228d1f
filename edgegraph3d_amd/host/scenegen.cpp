// Seeded synthetic workload generator: 3-D curves in a 400 mm cube, a
// sphere-cap camera rig, per-view sub-pixel polylines with node ids, SfM seeds with tracks,
// analytic fundamental matrices. The reference ships no usable input (example/dtu006/input.json
// is missing from the mount), so BASELINE.json's configs are realised with this.
// Not derived from any reference source file.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/eg3d_host.h"
#include "camera_model.hpp"

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {  // SplitMix64
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * uni(); }
  int irange(int a, int b) { return a + (int)(next() % (uint64_t)(b - a + 1)); }  // inclusive
  double normal() {
    double u1 = uni(), u2 = uni();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
  }
};

struct V3 {
  double x, y, z;
};
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 normalize(V3 a) {
  double n = std::sqrt(dot(a, a));
  return a * (1.0 / n);
}
static V3 rand_dir(Rng& r) {
  for (;;) {
    V3 v = {r.uni(-1, 1), r.uni(-1, 1), r.uni(-1, 1)};
    double n = dot(v, v);
    if (n > 1e-4 && n <= 1) return v * (1.0 / std::sqrt(n));
  }
}

struct Curve {
  int type;  // 0 line, 1 arc/helix, 2 full small circle (loop)
  V3 a, b;   // line endpoints
  V3 c, e1, e2, nrm;
  double rad, a0, a1, pitch;
  double length;
  V3 eval(double t) const {
    if (type == 0) return a + (b - a) * t;
    double ang = a0 + (a1 - a0) * t;
    return c + e1 * (rad * std::cos(ang)) + e2 * (rad * std::sin(ang)) + nrm * (pitch * t);
  }
};

struct Cam {
  float f, px, py;
  float R[9], C[3], t[3], P[16];
};

static bool project(const Cam& cam, V3 X, double& u, double& v) {
  const float* P = cam.P;
  double h0 = P[0] * X.x + P[1] * X.y + P[2] * X.z + P[3];
  double h1 = P[4] * X.x + P[5] * X.y + P[6] * X.z + P[7];
  double h2 = P[8] * X.x + P[9] * X.y + P[10] * X.z + P[11];
  if (h2 <= 1e-6) return false;
  u = h0 / h2;
  v = h1 / h2;
  return true;
}

}  // namespace

struct eg3d_synth {
  eg3d_synth_config cfg;
  std::vector<Cam> cams;
  std::vector<Curve> curves;
  std::vector<float> cam_P;
  std::vector<double> F;
  std::vector<uint8_t> F_valid;
  std::vector<uint32_t> view_pl_off, pl_vtx_off, pl_start, pl_end;
  std::vector<uint8_t> pl_valid;
  std::vector<uint32_t> pl_curve;  // generating 3-D curve of every polyline (for synthetic polyline match sets)
  uint32_t cur_curve = 0;
  std::vector<float> vtx_xy;
  std::vector<uint32_t> trk_off;
  std::vector<int32_t> trk_view;
  std::vector<float> trk_xy;
  std::vector<float> seed_truth;
  eg3d_scene scene;
  eg3d_seeds seeds;
  uint64_t total_segments;
};

extern "C" void eg3d_synth_default_config(eg3d_synth_config* c, int idx) {
  memset(c, 0, sizeof(*c));
  c->width = 1600;
  c->height = 1200;
  c->focal = 2890.f;
  c->ppx = 823.f;
  c->ppy = 619.f;
  c->obs_noise_px = 0.4f;
  c->vtx_noise_px = 0.15f;
  c->invalid_frac = 0.01f;
  c->seed_offset_px = 6.f;
  c->max_track = 12;
  c->rng_seed = 0xE63D2018ull + (uint64_t)idx;
  switch (idx) {
    case 2:  // C2: 8 views / 2k seeds / ~5k segments per view
      c->n_views = 8;
      c->n_seeds = 2000;
      c->n_curves = 82;
      break;
    case 3:  // C3': dtu006-shaped, 25 views / 6268 seeds / ~12-18k segments per view
      c->n_views = 25;
      c->n_seeds = 6268;
      c->n_curves = 225;
      break;
    case 4:  // C4: 200 views / 100k seeds / ~20k segments per view
      c->n_views = 200;
      c->n_seeds = 100000;
      c->n_curves = 295;
      break;
    case 5:  // C5 rig: 16 views so that k ~ U[3,10] of the 1 M-point filter workload (eg3d_synth_points) is real
      c->n_views = 16;
      c->n_seeds = 200;
      c->n_curves = 82;
      break;
    case 1:  // small: used by the CPU parity tests
      c->n_views = 6;
      c->n_seeds = 120;
      c->n_curves = 14;
      break;
    default:  // tiny
      c->n_views = 4;
      c->n_seeds = 40;
      c->n_curves = 8;
      break;
  }
}

static void make_cameras(eg3d_synth* s, Rng& rng) {
  const int V = s->cfg.n_views;
  s->cams.resize(V);
  const double cap = 55.0 * M_PI / 180.0;
  const double golden = M_PI * (3.0 - std::sqrt(5.0));
  for (int i = 0; i < V; i++) {
    double u = (i + 0.5) / V;
    double ct = 1.0 - u * (1.0 - std::cos(cap));
    double st = std::sqrt(std::max(0.0, 1.0 - ct * ct));
    double ph = golden * i;
    double r = rng.uni(600.0, 700.0);
    V3 C = {r * st * std::cos(ph), r * st * std::sin(ph), r * ct};
    V3 target = {rng.uni(-15, 15), rng.uni(-15, 15), rng.uni(-15, 15)};
    V3 z = normalize(target - C);
    V3 up = {0, 1, 0};
    if (std::fabs(dot(up, z)) > 0.95) up = {1, 0, 0};
    V3 x = normalize(cross(up, z));
    V3 y = cross(z, x);
    Cam& cam = s->cams[i];
    cam.f = s->cfg.focal;
    cam.px = s->cfg.ppx;
    cam.py = s->cfg.ppy;
    double Rd[9] = {x.x, x.y, x.z, y.x, y.y, y.z, z.x, z.y, z.z};
    for (int k = 0; k < 9; k++) cam.R[k] = (float)Rd[k];
    cam.C[0] = (float)C.x;
    cam.C[1] = (float)C.y;
    cam.C[2] = (float)C.z;
    eg3dh::translation_from_center(cam.R, cam.C, cam.t);
    eg3dh::camera_matrix(cam.f, cam.px, cam.py, cam.R, cam.t, cam.P);
  }
  s->cam_P.resize((size_t)V * 16);
  for (int i = 0; i < V; i++) memcpy(&s->cam_P[(size_t)i * 16], s->cams[i].P, sizeof(float) * 16);
  s->F.assign((size_t)V * V * 9, 0.0);
  s->F_valid.assign((size_t)V * V, 0);
  for (int i = 0; i < V; i++)
    for (int j = 0; j < V; j++)
      if (i != j) {
        const Cam& a = s->cams[i];
        const Cam& b = s->cams[j];
        eg3dh::fundamental_from_cameras(a.f, a.px, a.py, a.R, a.t, b.f, b.px, b.py, b.R, b.t,
                                        &s->F[((size_t)i * V + j) * 9]);
        s->F_valid[(size_t)i * V + j] = 1;
      }
}

static void make_curves(eg3d_synth* s, Rng& rng) {
  const int M = s->cfg.n_curves;
  s->curves.resize(M);
  for (int m = 0; m < M; m++) {
    Curve& c = s->curves[m];
    memset(&c, 0, sizeof(c));
    double u = rng.uni();
    if (u < 0.60) {
      c.type = 0;
      V3 mid = {rng.uni(-160, 160), rng.uni(-160, 160), rng.uni(-160, 160)};
      V3 d = rand_dir(rng);
      double len = rng.uni(100, 400);
      c.a = mid - d * (len * 0.5);
      c.b = mid + d * (len * 0.5);
      c.length = len;
    } else if (u < 0.94) {
      c.type = 1;
      c.c = {rng.uni(-140, 140), rng.uni(-140, 140), rng.uni(-140, 140)};
      c.nrm = rand_dir(rng);
      V3 t = rand_dir(rng);
      c.e1 = normalize(cross(c.nrm, t));
      c.e2 = cross(c.nrm, c.e1);
      c.rad = rng.uni(50, 200);
      c.a0 = rng.uni(0, 2 * M_PI);
      double span = rng.uni(60, 300) * M_PI / 180.0;
      c.a1 = c.a0 + span;
      c.pitch = rng.uni() < 0.4 ? rng.uni(-80, 80) : 0.0;
      c.length = std::sqrt((c.rad * span) * (c.rad * span) + c.pitch * c.pitch);
    } else {
      c.type = 2;
      c.c = {rng.uni(-150, 150), rng.uni(-150, 150), rng.uni(-150, 150)};
      c.nrm = rand_dir(rng);
      V3 t = rand_dir(rng);
      c.e1 = normalize(cross(c.nrm, t));
      c.e2 = cross(c.nrm, c.e1);
      c.rad = rng.uni(10, 20);
      c.a0 = rng.uni(0, 2 * M_PI);
      c.a1 = c.a0 + 2 * M_PI;
      c.pitch = 0;
      c.length = 2 * M_PI * c.rad;
    }
  }
}

static void emit_polyline(eg3d_synth* s, Rng& rng, const std::vector<float>& run, size_t v0, size_t v1, uint32_t n0,
                          uint32_t n1) {
  // vertices [v0, v1] inclusive of `run` (xy pairs)
  bool invalid = rng.uni() < s->cfg.invalid_frac;
  s->pl_start.push_back(n0);
  s->pl_end.push_back(n1);
  s->pl_curve.push_back(s->cur_curve);
  if (invalid) {
    s->pl_valid.push_back(0);
  } else {
    s->pl_valid.push_back(1);
    for (size_t i = v0; i <= v1; i++) {
      s->vtx_xy.push_back(run[2 * i]);
      s->vtx_xy.push_back(run[2 * i + 1]);
    }
    s->total_segments += (v1 - v0);
  }
  s->pl_vtx_off.push_back((uint32_t)(s->vtx_xy.size() / 2));
}

static void make_polylines(eg3d_synth* s, Rng& rng) {
  const int V = s->cfg.n_views;
  const double W = s->cfg.width, H = s->cfg.height, margin = 3.0;
  s->view_pl_off.assign(1, 0);
  s->pl_vtx_off.assign(1, 0);
  s->total_segments = 0;
  for (int v = 0; v < V; v++) {
    uint32_t next_node = 0;
    const Cam& cam = s->cams[v];
    for (const Curve& c : s->curves) {
      s->cur_curve = (uint32_t)(&c - s->curves.data());
      int ns = std::max(8, (int)std::ceil(c.length / 0.2));
      std::vector<float> run;
      bool in_run = false, whole_visible = true;
      double lu = 0, lv = 0, acc = 0, want = 0, pu = 0, pv = 0;
      auto flush = [&](bool closed_loop) {
        size_t nv = run.size() / 2;
        if (nv >= 2) {
          if (closed_loop && nv >= 4 && nv <= 59) {
            // close the loop: repeat the first vertex, single polyline, start == end (Q8)
            run.push_back(run[0]);
            run.push_back(run[1]);
            uint32_t n0 = next_node++;
            emit_polyline(s, rng, run, 0, nv, n0, n0);
          } else {
            size_t pos = 0;
            uint32_t n0 = next_node++;
            while (pos < nv - 1) {
              size_t want_n = (size_t)rng.irange(5, 60);
              size_t last = std::min(nv - 1, pos + want_n - 1);
              if (nv - 1 - last < 2) last = nv - 1;  // do not leave a 1-segment crumb
              uint32_t n1 = next_node++;
              emit_polyline(s, rng, run, pos, last, n0, n1);
              n0 = n1;
              pos = last;
            }
          }
        }
        run.clear();
      };
      for (int i = 0; i <= ns; i++) {
        double t = (double)i / ns;
        V3 X = c.eval(t);
        double uu, vv;
        bool vis = project(cam, X, uu, vv) && uu > margin && uu < W - margin && vv > margin && vv < H - margin;
        if (!vis) {
          whole_visible = false;
          if (in_run) {
            // close the run at the last visible sample
            if (std::hypot(pu - lu, pv - lv) > 2.0) {
              run.push_back((float)(pu + rng.normal() * s->cfg.vtx_noise_px));
              run.push_back((float)(pv + rng.normal() * s->cfg.vtx_noise_px));
            }
            flush(false);
            in_run = false;
          }
          continue;
        }
        if (!in_run) {
          in_run = true;
          run.push_back((float)(uu + rng.normal() * s->cfg.vtx_noise_px));
          run.push_back((float)(vv + rng.normal() * s->cfg.vtx_noise_px));
          lu = uu;
          lv = vv;
          acc = 0;
          want = rng.uni(8, 12);
        } else {
          acc += std::hypot(uu - pu, vv - pv);
          if (acc >= want) {
            run.push_back((float)(uu + rng.normal() * s->cfg.vtx_noise_px));
            run.push_back((float)(vv + rng.normal() * s->cfg.vtx_noise_px));
            lu = uu;
            lv = vv;
            acc = 0;
            want = rng.uni(8, 12);
          }
        }
        pu = uu;
        pv = vv;
      }
      if (in_run) {
        bool loop = (c.type == 2) && whole_visible;
        if (!loop && std::hypot(pu - lu, pv - lv) > 2.0) {
          run.push_back((float)(pu + rng.normal() * s->cfg.vtx_noise_px));
          run.push_back((float)(pv + rng.normal() * s->cfg.vtx_noise_px));
        }
        flush(loop);
      }
    }
    s->view_pl_off.push_back((uint32_t)s->pl_start.size());
  }
}

static V3 curve_point(const eg3d_synth* s, Rng& rng, double& total_len_cache, std::vector<double>& cdf) {
  if (cdf.empty()) {
    double acc = 0;
    for (const Curve& c : s->curves) {
      acc += c.length;
      cdf.push_back(acc);
    }
    total_len_cache = acc;
  }
  double u = rng.uni() * total_len_cache;
  size_t ci = std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin();
  if (ci >= s->curves.size()) ci = s->curves.size() - 1;
  return s->curves[ci].eval(rng.uni());
}

static void make_seeds(eg3d_synth* s, Rng& rng) {
  const int V = s->cfg.n_views;
  const double W = s->cfg.width, H = s->cfg.height;
  s->trk_off.assign(1, 0);
  double total = 0;
  std::vector<double> cdf;
  std::vector<int> vis;
  std::vector<double> pu(V), pv(V);
  for (uint32_t i = 0; i < s->cfg.n_seeds; i++) {
    for (int attempt = 0;; attempt++) {
      V3 T = curve_point(s, rng, total, cdf);
      double r_px = rng.uni(0, s->cfg.seed_offset_px);
      V3 Q = T + rand_dir(rng) * (r_px * 650.0 / s->cfg.focal);
      vis.clear();
      for (int v = 0; v < V; v++) {
        double uu, vv;
        if (project(s->cams[v], Q, uu, vv) && uu > 12 && uu < W - 12 && vv > 12 && vv < H - 12) {
          vis.push_back(v);
          pu[v] = uu;
          pv[v] = vv;
        }
      }
      if ((int)vis.size() < 3 && attempt < 1000) continue;
      int kmax = std::min((int)vis.size(), s->cfg.max_track);
      int k = kmax >= 3 ? rng.irange(3, kmax) : (int)vis.size();
      for (int a = 0; a < k; a++) {
        int b = rng.irange(a, (int)vis.size() - 1);
        std::swap(vis[a], vis[b]);
      }
      std::sort(vis.begin(), vis.begin() + k);
      for (int a = 0; a < k; a++) {
        int v = vis[a];
        s->trk_view.push_back(v);
        s->trk_xy.push_back((float)(pu[v] + rng.normal() * s->cfg.obs_noise_px));
        s->trk_xy.push_back((float)(pv[v] + rng.normal() * s->cfg.obs_noise_px));
      }
      s->trk_off.push_back((uint32_t)s->trk_view.size());
      s->seed_truth.push_back((float)T.x);
      s->seed_truth.push_back((float)T.y);
      s->seed_truth.push_back((float)T.z);
      break;
    }
  }
}

extern "C" eg3d_synth* eg3d_synth_create(const eg3d_synth_config* cfg) {
  if (!cfg || cfg->n_views < 2) return nullptr;
  eg3d_synth* s = new eg3d_synth();
  s->cfg = *cfg;
  Rng rng(cfg->rng_seed);
  make_cameras(s, rng);
  make_curves(s, rng);
  make_polylines(s, rng);
  make_seeds(s, rng);
  eg3d_scene& sc = s->scene;
  sc.n_views = cfg->n_views;
  sc.width = cfg->width;
  sc.height = cfg->height;
  sc.cam_P = s->cam_P.data();
  sc.F = s->F.data();
  sc.F_valid = s->F_valid.data();
  sc.view_pl_off = s->view_pl_off.data();
  sc.pl_vtx_off = s->pl_vtx_off.data();
  sc.vtx_xy = s->vtx_xy.data();
  sc.pl_start = s->pl_start.data();
  sc.pl_end = s->pl_end.data();
  sc.pl_valid = s->pl_valid.data();
  s->seeds.n_seeds = cfg->n_seeds;
  s->seeds.trk_off = s->trk_off.data();
  s->seeds.trk_view = s->trk_view.data();
  s->seeds.trk_xy = s->trk_xy.data();
  return s;
}

extern "C" const eg3d_scene* eg3d_synth_scene(const eg3d_synth* s) { return &s->scene; }
extern "C" const eg3d_seeds* eg3d_synth_seeds(const eg3d_synth* s) { return &s->seeds; }
extern "C" const float* eg3d_synth_seed_truth(const eg3d_synth* s) { return s->seed_truth.data(); }
extern "C" const uint32_t* eg3d_synth_polyline_curve(const eg3d_synth* s) { return s->pl_curve.data(); }
extern "C" int eg3d_synth_n_curves(const eg3d_synth* s) { return (int)s->curves.size(); }
extern "C" int eg3d_synth_camera(const eg3d_synth* s, int view, float* focal, float* ppx, float* ppy, float* R9,
                                 float* C3) {
  if (!s || view < 0 || view >= (int)s->cams.size()) return -1;
  const Cam& c = s->cams[view];
  *focal = c.f;
  *ppx = c.px;
  *ppy = c.py;
  memcpy(R9, c.R, sizeof(float) * 9);
  memcpy(C3, c.C, sizeof(float) * 3);
  return 0;
}
extern "C" uint64_t eg3d_synth_total_segments(const eg3d_synth* s) { return s->total_segments; }
extern "C" void eg3d_synth_destroy(eg3d_synth* s) { delete s; }
extern "C" void eg3d_host_free(void* p) { free(p); }

extern "C" int eg3d_synth_points(const eg3d_synth* s, uint64_t n_points, uint64_t rng_seed, float** X,
                                 uint32_t** obs_off, int32_t** obs_view, float** obs_xy) {
  if (!s) return -1;
  Rng rng(rng_seed);
  const int V = s->cfg.n_views;
  const double W = s->cfg.width, H = s->cfg.height;
  std::vector<float> vX, vxy;
  std::vector<uint32_t> voff(1, 0);
  std::vector<int32_t> vview;
  vX.reserve(n_points * 3);
  double total = 0;
  std::vector<double> cdf;
  std::vector<int> vis;
  std::vector<double> pu(V), pv(V);
  for (uint64_t i = 0; i < n_points; i++) {
    for (int attempt = 0;; attempt++) {
      V3 T = curve_point(s, rng, total, cdf);
      vis.clear();
      for (int v = 0; v < V; v++) {
        double uu, vv;
        if (project(s->cams[v], T, uu, vv) && uu > 5 && uu < W - 5 && vv > 5 && vv < H - 5) {
          vis.push_back(v);
          pu[v] = uu;
          pv[v] = vv;
        }
      }
      if ((int)vis.size() < 3 && attempt < 1000) continue;
      int kmax = std::min((int)vis.size(), 10);
      int k = kmax >= 3 ? rng.irange(3, kmax) : (int)vis.size();
      for (int a = 0; a < k; a++) {
        int b = rng.irange(a, (int)vis.size() - 1);
        std::swap(vis[a], vis[b]);
      }
      std::sort(vis.begin(), vis.begin() + k);
      bool gross = rng.uni() < 0.05;
      int gross_at = gross ? rng.irange(0, k - 1) : -1;
      for (int a = 0; a < k; a++) {
        int v = vis[a];
        double du = rng.normal() * 0.5, dv = rng.normal() * 0.5;
        if (a == gross_at) {
          double r = rng.uni(20, 50), ang = rng.uni(0, 2 * M_PI);
          du += r * std::cos(ang);
          dv += r * std::sin(ang);
        }
        vview.push_back(v);
        vxy.push_back((float)(pu[v] + du));
        vxy.push_back((float)(pv[v] + dv));
      }
      voff.push_back((uint32_t)vview.size());
      vX.push_back((float)(T.x + rng.normal() * 2.0));
      vX.push_back((float)(T.y + rng.normal() * 2.0));
      vX.push_back((float)(T.z + rng.normal() * 2.0));
      break;
    }
  }
  *X = (float*)malloc(sizeof(float) * vX.size());
  memcpy(*X, vX.data(), sizeof(float) * vX.size());
  *obs_off = (uint32_t*)malloc(sizeof(uint32_t) * voff.size());
  memcpy(*obs_off, voff.data(), sizeof(uint32_t) * voff.size());
  *obs_view = (int32_t*)malloc(sizeof(int32_t) * std::max<size_t>(1, vview.size()));
  memcpy(*obs_view, vview.data(), sizeof(int32_t) * vview.size());
  *obs_xy = (float*)malloc(sizeof(float) * std::max<size_t>(1, vxy.size()));
  memcpy(*obs_xy, vxy.data(), sizeof(float) * vxy.size());
  return 0;
}
