// eg3d_dev_geom.h — 2-D geometry and polyline-walking primitives of the MI355X path.
//
// These are the per-lane building blocks of the gfx950 kernels in eg3d_kernels.hip. They are
// written against flat device arrays (struct-of-arrays scene, CSR polylines) instead of the
// reference's vector-of-struct graph. Each function names the reference behaviour it must
// reproduce (file:line relative to the reference tree) — behaviour, not code: the layout,
// control flow and data types here are this project's.
//
// Arithmetic contract: compiled with -ffp-contract=off; float/double mix and
// evaluation order fixed so results are bit-identical to the CPU oracle.
//
// EG3D_HD expands to __host__ __device__ under hipcc. The host instantiation is used by the
// grid builder (host/grid_build.cpp) and by the test-only host simulation of the kernels
// (tests/hostsim); the product's compute path runs these on the GPU only.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define EG3D_HD __host__ __device__ inline
// the large per-chain routines must stay inlined into the expand kernel: when the inliner outlines
// one of them (it did, depending on unrelated code-size changes) the call ABI costs a third of the
// kernel's speed (more scratch, registers saved around the call)
#define EG3D_HD_FLAT __host__ __device__ inline __attribute__((always_inline))
#else
#define EG3D_HD inline
#define EG3D_HD_FLAT inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// __fsqrt_rn lowers to the approximate native sqrt on gfx950 (measured: 15 % of results are one
// ulp off); __builtin_sqrtf under -fhip-fp32-correctly-rounded-divide-sqrt is correctly rounded
// (measured bit-identical to x86 sqrtf on 2e5 random inputs, tests/test_gpu_arith.py).
#define EG3D_SQRTF(x) __builtin_sqrtf(x)
#define EG3D_SQRT(x) __dsqrt_rn(x)
#else
#include <math.h>
#define EG3D_SQRTF(x) sqrtf(x)
#define EG3D_SQRT(x) sqrt(x)
#endif

#ifndef EG3D_STAT
#define EG3D_STAT(i) /* call counters of the host simulation (tests/hostsim); nothing on the device */
#endif

namespace eg3d {

struct f2 {
  float x, y;
};

// A polyline as the kernels see it: a contiguous slice of the vertex array + its two node ids.
// VPtr is `const f2*` everywhere except in the expand kernel's side walks, which walk a copy of
// the polyline staged in LDS through an address_space(3) pointer (ds_read instead of flat loads).
template <class VPtr>
struct PlRefT {
  VPtr v;
  uint32_t n;      // vertex count
  uint32_t start;  // node ids (reference polyline::start / ::end)
  uint32_t end;
  const float* bb = nullptr;  // optional: bounding boxes (min x, min y, max x, max y) of blocks of EG3D_BB_SEGS segments
};
#define EG3D_BB_SEGS 8
typedef PlRefT<const f2*> PlRef;

// A point on a polyline: segment index + coordinates (reference pl_point).
struct PlPt {
  uint32_t seg;
  float x, y;
};

// walk status bits
enum : uint32_t {
  WALK_FOUND = 1u,
  WALK_EXTREME = 2u,        // reached the polyline extreme
  WALK_QUASIPARALLEL = 4u,  // stopped on a quasi-parallel segment
  WALK_BOUND = 8u,          // hit found but outside [min,max] distance
  WALK_BAD_DIR = 16u        // direction is neither end of this polyline (Q15)
};

// |a-b|^2: differences in float, squares and sum in double, one rounding to float
// (geometric_utilities.cpp:555-557, Q5).
EG3D_HD float dist2(float ax, float ay, float bx, float by) {
  double dx = (double)(ax - bx);
  double dy = (double)(ay - by);
  return (float)(dx * dx + dy * dy);
}
EG3D_HD float dist(float ax, float ay, float bx, float by) { return EG3D_SQRTF(dist2(ax, ay, bx, by)); }

EG3D_HD float dotf(float ax, float ay, float bx, float by) {
  float p = ax * bx;
  float q = ay * by;
  return p + q;
}

// a + ratio*(b-a), per component (geometric_utilities.cpp:1370-1372)
EG3D_HD void lerp_from(float ax, float ay, float bx, float by, float ratio, float& ox, float& oy) {
  float dx = bx - ax, dy = by - ay;
  float rx = ratio * dx, ry = ratio * dy;
  ox = ax + rx;
  oy = ay + ry;
}

// Closest point of segment vw to p; returns squared distance (geometric_utilities.cpp:940-954).
EG3D_HD float seg_closest(float px, float py, float vx, float vy, float wx, float wy, float& qx, float& qy) {
  const float l2 = dist2(vx, vy, wx, wy);
  if (l2 == 0.0f) {
    qx = vx;
    qy = vy;
    return dist2(px, py, vx, vy);
  }
  const float q = dotf(px - vx, py - vy, wx - vx, wy - vy) / l2;
  const float m = (q < 1.0f) ? q : 1.0f;  // std::min(1,q): NaN -> 1
  const float t = (0.0f < m) ? m : 0.0f;  // std::max(0,m)
  float ex = wx - vx, ey = wy - vy;
  float tx = t * ex, ty = t * ey;
  qx = vx + tx;
  qy = vy + ty;
  return dist2(px, py, qx, qy);
}

// Segment (x1,y1)-(x2,y2) against line (a,b,c): returns true and the hit when 0<=t<=1
// (geometric_utilities.cpp:272-312).
EG3D_HD bool seg_line_hit(float x1, float y1, float x2, float y2, float la, float lb, float lc, float& hx,
                          float& hy) {
  float dx = x2 - x1, dy = y2 - y1;
  float n0 = la * x1, n1 = lb * y1;
  float num = (n0 + n1) + lc;
  float d0 = la * dx, d1 = lb * dy;
  float den = d0 + d1;
  if (den != 0.0f) {
    float t = -num / den;
    if (t >= 0.0f && t <= 1.0f) {
      float tx = t * dx, ty = t * dy;
      hx = x1 + tx;
      hy = y1 + ty;
      return true;
    }
  }
  return false;
}

EG3D_HD float point_line_dist(float px, float py, float la, float lb, float lc) {
  float t0 = la * px, t1 = lb * py;
  float den = (t0 + t1) + lc;
  den *= den;
  float a2 = la * la, b2 = lb * lb;
  return EG3D_SQRTF(den / (a2 + b2));
}

// Signed cosine between the segment (walking order) and the line direction (1,-a/b) or (0,1)
// (geometric_utilities.cpp:590-618, Q14).
EG3D_HD float seg_line_cos(float x1, float y1, float x2, float y2, float la, float lb) {
  float ax = x2 - x1, ay = y2 - y1;
  float bx, by;
  if (lb == 0.0f) {
    bx = 0.0f;
    by = 1.0f;
  } else {
    bx = 1.0f;
    by = -la / lb;
  }
  float d = dotf(ax, ay, bx, by);
  float aa = dotf(ax, ay, ax, ay);
  float bb = dotf(bx, by, bx, by);
  return d / EG3D_SQRTF(aa * bb);
}

// Direction (1,-a/b) or (0,1) of a line and its squared norm: invariant along a walk, so the walk
// evaluates the division once (same operands => same bits as per-segment evaluation).
struct LineDir {
  float bx, by, bb;
};
EG3D_HD LineDir line_dir(float la, float lb) {
  LineDir d;
  if (lb == 0.0f) {
    d.bx = 0.0f;
    d.by = 1.0f;
  } else {
    d.bx = 1.0f;
    d.by = -la / lb;
  }
  d.bb = dotf(d.bx, d.by, d.bx, d.by);
  return d;
}
// seg_line_cos(...) > thr, decided on the squares when that is unambiguous (relative margin 1e-4
// >> the few-ulp error of either form); the division and square root are evaluated only in the
// knife-edge band and for degenerate magnitudes — same decisions as the plain expression.
EG3D_HD bool seg_line_cos_gt(float x1, float y1, float x2, float y2, const LineDir& ld, float thr) {
  float ax = x2 - x1, ay = y2 - y1;
  float d = dotf(ax, ay, ld.bx, ld.by);
  float aa = dotf(ax, ay, ax, ay);
  float prod = aa * ld.bb;
  if (prod > 1e-30f && prod < 1e30f) {
    if (d <= 0.0f) return false;
    const float d2 = d * d, rhs = (thr * thr) * prod;
    if (d2 > rhs * 1.0001f) return true;
    if (d2 < rhs * 0.9999f) return false;
  }
  return d / EG3D_SQRTF(prod) > thr;
}

// Segment/line test with the quasi-parallel guard (cos > 0.965 within 5 px),
// geometric_utilities.cpp:365-430. Returns bit0 = hit found, bit1 = quasi-parallel within distance.
EG3D_HD uint32_t seg_line_hit_guarded(float x1, float y1, float x2, float y2, float la, float lb, float lc,
                                      const LineDir& ld, float& hx, float& hy) {
  const float QP_COS = (float)0.965;
  const float QP_DIST = 5.0f;
  uint32_t r = 0;
  float dx = x2 - x1, dy = y2 - y1;
  float n0 = la * x1, n1 = lb * y1;
  float num = (n0 + n1) + lc;
  float d0 = la * dx, d1 = lb * dy;
  float den = d0 + d1;
  if (den != 0.0f) {
    float t = -num / den;
    if (t >= 0.0f && t <= 1.0f) {
      float tx = t * dx, ty = t * dy;
      hx = x1 + tx;
      hy = y1 + ty;
      r |= 1u;
    }
    if (seg_line_cos_gt(x1, y1, x2, y2, ld, QP_COS)) {
      float distance;
      if (t < 0.0f)
        distance = point_line_dist(x1, y1, la, lb, lc);
      else if (t > 1.0f)
        distance = point_line_dist(x2, y2, la, lb, lc);
      else
        distance = 0.0f;
      if (distance <= QP_DIST) r |= 2u;
    }
  } else {
    float distance = point_line_dist(x1, y1, la, lb, lc);
    if (distance <= QP_DIST) r |= 2u;
  }
  return r;
}

// ---------------------------------------------------------------- walking -----
// Next point at `distance` (Euclidean from p, not arc length) towards node `direction`
// (polyline_graph_2d.cpp:391-447). Returns WALK_FOUND, WALK_EXTREME or WALK_BAD_DIR.
EG3D_HD uint32_t walk_by_distance(const PlRef& pl, const PlPt& p, uint32_t direction, float distance, PlPt& out) {
  float prevdist = 0.0f, curdist, ratio;
  if (direction == pl.start) {
    curdist = dist(pl.v[p.seg].x, pl.v[p.seg].y, p.x, p.y);
    if (curdist >= distance) {
      ratio = distance / curdist;
      out.seg = p.seg;
      lerp_from(p.x, p.y, pl.v[p.seg].x, pl.v[p.seg].y, ratio, out.x, out.y);
      return WALK_FOUND;
    }
    uint32_t i;
    for (i = p.seg; i > 0; i--) {
      prevdist = curdist;
      curdist = dist(pl.v[i - 1].x, pl.v[i - 1].y, p.x, p.y);
      if (curdist >= distance) break;
    }
    if (i == 0) {
      out.seg = 0;
      out.x = pl.v[0].x;
      out.y = pl.v[0].y;
      return WALK_EXTREME;
    }
    ratio = (distance - prevdist) / (curdist - prevdist);
    out.seg = i - 1;
    lerp_from(pl.v[i].x, pl.v[i].y, pl.v[i - 1].x, pl.v[i - 1].y, ratio, out.x, out.y);
    return WALK_FOUND;
  } else if (direction == pl.end) {
    const uint32_t n = pl.n;
    if (p.seg >= n - 1) {
      out.seg = n - 2;
      out.x = pl.v[n - 1].x;
      out.y = pl.v[n - 1].y;
      return WALK_EXTREME;
    }
    curdist = dist(pl.v[p.seg + 1].x, pl.v[p.seg + 1].y, p.x, p.y);
    if (curdist >= distance) {
      ratio = distance / curdist;
      out.seg = p.seg;
      lerp_from(p.x, p.y, pl.v[p.seg + 1].x, pl.v[p.seg + 1].y, ratio, out.x, out.y);
      return WALK_FOUND;
    }
    uint32_t i;
    for (i = p.seg + 1; i < n - 1; i++) {
      prevdist = curdist;
      curdist = dist(pl.v[i + 1].x, pl.v[i + 1].y, p.x, p.y);
      if (curdist >= distance) break;
    }
    if (i == n - 1) {
      out.seg = n - 2;
      out.x = pl.v[n - 1].x;
      out.y = pl.v[n - 1].y;
      return WALK_EXTREME;
    }
    ratio = (distance - prevdist) / (curdist - prevdist);
    out.seg = i;
    lerp_from(pl.v[i].x, pl.v[i].y, pl.v[i + 1].x, pl.v[i + 1].y, ratio, out.x, out.y);
    return WALK_FOUND;
  }
  out = p;
  return WALK_EXTREME | WALK_BAD_DIR;  // Q15: undefined in the reference; the walk fails
}

// Next intersection with the line towards `direction`, stopping at quasi-parallel segments;
// optional [min,max] distance window (polyline_graph_2d.cpp:579-655 and :657-780).
template <class VPtr>
EG3D_HD uint32_t walk_by_line(const PlRefT<VPtr>& pl, const PlPt& p, uint32_t direction, float la, float lb, float lc,
                              bool bounded, float min_d, float max_d, PlPt& out) {
  float hx = 0.0f, hy = 0.0f;
  uint32_t r;
  uint32_t seg_found = 0;
  bool got = false;
  const LineDir ld = line_dir(la, lb);
  EG3D_STAT(5);
  if (direction == pl.start) {
    r = seg_line_hit_guarded(p.x, p.y, pl.v[p.seg].x, pl.v[p.seg].y, la, lb, lc, ld, hx, hy);
    if (r & 2u) return WALK_QUASIPARALLEL;
    if (r & 1u) {
      got = true;
      seg_found = p.seg;
    } else {
      for (uint32_t i = p.seg; i > 0; i--) {
        EG3D_STAT(6);
        r = seg_line_hit_guarded(pl.v[i].x, pl.v[i].y, pl.v[i - 1].x, pl.v[i - 1].y, la, lb, lc, ld, hx, hy);
        if (r & 2u) return WALK_QUASIPARALLEL;
        if (r & 1u) {
          got = true;
          seg_found = i - 1;
          break;
        }
      }
      if (!got) return WALK_EXTREME;
    }
  } else if (direction == pl.end) {
    r = seg_line_hit_guarded(p.x, p.y, pl.v[p.seg + 1].x, pl.v[p.seg + 1].y, la, lb, lc, ld, hx, hy);
    if (r & 2u) return WALK_QUASIPARALLEL;
    if (r & 1u) {
      got = true;
      seg_found = p.seg;
    } else {
      for (uint32_t i = p.seg + 1; i < pl.n - 1; i++) {
        EG3D_STAT(6);
        r = seg_line_hit_guarded(pl.v[i].x, pl.v[i].y, pl.v[i + 1].x, pl.v[i + 1].y, la, lb, lc, ld, hx, hy);
        if (r & 2u) return WALK_QUASIPARALLEL;
        if (r & 1u) {
          got = true;
          seg_found = i;
          break;
        }
      }
      if (!got) return WALK_EXTREME;
    }
  } else {
    return WALK_BAD_DIR;  // Q15: the reference leaves every flag false
  }
  out.seg = seg_found;
  out.x = hx;
  out.y = hy;
  if (bounded) {
    float dsq = dist2(hx, hy, p.x, p.y);
    if (dsq < (min_d * min_d) || dsq > (max_d * max_d)) return WALK_BOUND;
  }
  return WALK_FOUND;
}

// The two walks with their vertex loads IN FLIGHT: the plain loops above fetch one vertex per segment, each a
// dependent trip to memory (what a lane of the K3a engine spends its time on: a walk is ~5 segments of ~40
// instructions, a trip ~1000 cycles). These keep a window of five vertices loaded ahead of the segment being tested
// (indices clamped to the polyline), so a walk costs about one trip plus a fifth per segment. Same tests in the
// same order on the same operands: same result, bit for bit (tests/test_cpu_parity.py compares them with the plain
// walks on random polylines, both directions, every start segment).
template <class VPtr>
struct VtxWindow {
  VPtr v;
  int32_t first, stepv, hi;
  f2 w0, w1, w2, w3, w4;
  EG3D_HD f2 at(int32_t k) const {
    int32_t j = first + stepv * k;
    j = j < 0 ? 0 : (j > hi ? hi : j);
    return v[j];
  }
  EG3D_HD void open(VPtr v_, int32_t first_, int32_t stepv_, int32_t hi_) {
    v = v_;
    first = first_;
    stepv = stepv_;
    hi = hi_;
    w0 = at(0);
    w1 = at(1);
    w2 = at(2);
    w3 = at(3);
    w4 = at(4);
  }
  EG3D_HD void shift(int32_t k) {  // after the test of (u[k-1], u[k]): w0 becomes u[k]
    w0 = w1;
    w1 = w2;
    w2 = w3;
    w3 = w4;
    w4 = at(k + 4);
  }
};

template <class VPtr>
EG3D_HD uint32_t walk_by_distance_pf(const PlRefT<VPtr>& pl, const PlPt& p, uint32_t direction, float distance, PlPt& out) {
  const bool to_start = direction == pl.start;
  if (!to_start && direction != pl.end) {
    out = p;
    return WALK_EXTREME | WALK_BAD_DIR;
  }
  const int32_t n = (int32_t)pl.n;
  if (!to_start && (int32_t)p.seg >= n - 1) {
    out.seg = (uint32_t)(n - 2);
    out.x = pl.v[n - 1].x;
    out.y = pl.v[n - 1].y;
    return WALK_EXTREME;
  }
  const int32_t first = to_start ? (int32_t)p.seg : (int32_t)p.seg + 1;
  const int32_t count = to_start ? first + 1 : n - first;  // vertices ahead: u[0] = v[first], ... up to the extreme
  VtxWindow<VPtr> W;
  W.open(pl.v, first, to_start ? -1 : 1, n - 1);
  float prevdist = 0.0f, curdist = dist(W.w0.x, W.w0.y, p.x, p.y), ratio;
  if (curdist >= distance) {
    ratio = distance / curdist;
    out.seg = p.seg;
    lerp_from(p.x, p.y, W.w0.x, W.w0.y, ratio, out.x, out.y);
    return WALK_FOUND;
  }
  int32_t k;
  for (k = 1; k < count; k++) {
    prevdist = curdist;
    curdist = dist(W.w1.x, W.w1.y, p.x, p.y);
    if (curdist >= distance) break;
    W.shift(k);
  }
  if (k >= count) {
    const int32_t e = to_start ? 0 : n - 1;
    out.seg = to_start ? 0u : (uint32_t)(n - 2);
    out.x = pl.v[e].x;
    out.y = pl.v[e].y;
    return WALK_EXTREME;
  }
  ratio = (distance - prevdist) / (curdist - prevdist);
  out.seg = (uint32_t)(to_start ? first - k : first + k - 1);
  lerp_from(W.w0.x, W.w0.y, W.w1.x, W.w1.y, ratio, out.x, out.y);
  return WALK_FOUND;
}

// walk_by_line_pf in two halves, so that a caller with several walks to make can OPEN all their windows first (the
// vertex requests of all of them in flight together) and then run them: open = request the five vertices ahead of the
// position; run = the tests. walk_by_line_pf is the two back to back.
template <class VPtr>
EG3D_HD bool walk_by_line_open(const PlRefT<VPtr>& pl, const PlPt& p, uint32_t direction, VtxWindow<VPtr>& W) {
  const bool to_start = direction == pl.start;
  if (!to_start && direction != pl.end) return false;  // Q15: the reference leaves every flag false
  const int32_t first = to_start ? (int32_t)p.seg : (int32_t)p.seg + 1;
  W.open(pl.v, first, to_start ? -1 : 1, (int32_t)pl.n - 1);
  return true;
}
template <class VPtr>
EG3D_HD uint32_t walk_by_line_run(const PlRefT<VPtr>& pl, const PlPt& p, uint32_t direction, float la, float lb, float lc,
                                  bool bounded, float min_d, float max_d, VtxWindow<VPtr>& W, PlPt& out) {
  const bool to_start = direction == pl.start;
  const LineDir ld = line_dir(la, lb);
  const int32_t n = (int32_t)pl.n;
  const int32_t first = to_start ? (int32_t)p.seg : (int32_t)p.seg + 1;
  const int32_t count = to_start ? first + 1 : n - first;
  float hx = 0.0f, hy = 0.0f;
  uint32_t seg_found = p.seg;
  uint32_t r = seg_line_hit_guarded(p.x, p.y, W.w0.x, W.w0.y, la, lb, lc, ld, hx, hy);
  if (r & 2u) return WALK_QUASIPARALLEL;
  if (!(r & 1u)) {
    bool got = false;
    for (int32_t k = 1; k < count; k++) {
      r = seg_line_hit_guarded(W.w0.x, W.w0.y, W.w1.x, W.w1.y, la, lb, lc, ld, hx, hy);
      if (r & 2u) return WALK_QUASIPARALLEL;
      if (r & 1u) {
        got = true;
        seg_found = (uint32_t)(to_start ? first - k : first + k - 1);
        break;
      }
      W.shift(k);
    }
    if (!got) return WALK_EXTREME;
  }
  out.seg = seg_found;
  out.x = hx;
  out.y = hy;
  if (bounded) {
    float dsq = dist2(hx, hy, p.x, p.y);
    if (dsq < (min_d * min_d) || dsq > (max_d * max_d)) return WALK_BOUND;
  }
  return WALK_FOUND;
}
template <class VPtr>
EG3D_HD uint32_t walk_by_line_pf(const PlRefT<VPtr>& pl, const PlPt& p, uint32_t direction, float la, float lb, float lc,
                                 bool bounded, float min_d, float max_d, PlPt& out) {
  VtxWindow<VPtr> W;
  if (!walk_by_line_open(pl, p, direction, W)) return WALK_BAD_DIR;
  return walk_by_line_run(pl, p, direction, la, lb, lc, bounded, min_d, max_d, W, out);
}

// Closest point of a whole polyline, first minimal segment wins (polyline_graph_2d.cpp:845-862).
EG3D_HD float polyline_closest(const PlRef& pl, float px, float py, PlPt& out) {
  float bx, by;
  float best = seg_closest(px, py, pl.v[0].x, pl.v[0].y, pl.v[1].x, pl.v[1].y, bx, by);
  uint32_t bseg = 0;
  for (uint32_t i = 2; i < pl.n; i++) {
    float cx, cy;
    float d = seg_closest(px, py, pl.v[i - 1].x, pl.v[i - 1].y, pl.v[i].x, pl.v[i].y, cx, cy);
    if (d < best) {
      best = d;
      bx = cx;
      by = cy;
      bseg = i - 1;
    }
  }
  out.seg = bseg;
  out.x = bx;
  out.y = by;
  return best;
}

// The same over the segments [s0, s1) only (s1 <= n-1): the range that starts at segment 0 takes
// that segment as its initial best exactly like polyline_closest; later ranges start from +inf, so
// that merging the ranges' results by (smaller distance, then smaller segment index) reproduces the
// whole-polyline scan. Returns +inf with seg = 0xffffffff for an empty range.
EG3D_HD float polyline_closest_range(const PlRef& pl, float px, float py, uint32_t s0, uint32_t s1, PlPt& out) {
  float bx = 0.0f, by = 0.0f;
  float best = __builtin_inff();
  uint32_t bseg = 0xffffffffu;
  uint32_t i = s0;
  if (s0 == 0 && s1 > 0) {
    best = seg_closest(px, py, pl.v[0].x, pl.v[0].y, pl.v[1].x, pl.v[1].y, bx, by);
    bseg = 0;
    i = 1;
  }
  for (; i < s1; i++) {
    float cx, cy;
    float d = seg_closest(px, py, pl.v[i].x, pl.v[i].y, pl.v[i + 1].x, pl.v[i + 1].y, cx, cy);
    if (d < best) {
      best = d;
      bx = cx;
      by = cy;
      bseg = i;
    }
  }
  out.seg = bseg;
  out.x = bx;
  out.y = by;
  return best;
}

// polyline_closest_range with a PRE-TEST per block of EG3D_BB_SEGS segments: a block whose bounding box is farther
// from p than the best distance found so far cannot hold the closest segment and is not scanned. The result is the
// one of the plain scan, bit for bit: (1) segment 0 is the initial best exactly as there (NaN included: a NaN best
// is never replaced, and `bound > NaN` never prunes); (2) the block nearest to p is scanned next, which makes the
// bound tight at once; (3) then every block in ascending order — a segment replaces the best when it is strictly
// closer, or EQUALLY close with a smaller index (the plain scan keeps the first of equal distances). The bound is
// conservative in floating point: the computed distance of a segment is at least (gap - E)^2 (1 - 1e-6) per axis,
// E = 8 ulp of the largest coordinate involved (the closest point is v + t (w - v), t in [0,1], rounded; the
// difference p - q is rounded once; the squares are formed in double), and a NaN / infinite p never prunes.
EG3D_HD float polyline_closest_pruned(const PlRef& pl, float px, float py, uint32_t s0, uint32_t s1, PlPt& out) {
  if (!pl.bb || s1 <= s0) return polyline_closest_range(pl, px, py, s0, s1, out);
  float bx = 0.0f, by = 0.0f;
  float best = __builtin_inff();
  uint32_t bseg = 0xffffffffu;
  uint32_t first = s0;
  if (s0 == 0) {
    best = seg_closest(px, py, pl.v[0].x, pl.v[0].y, pl.v[1].x, pl.v[1].y, bx, by);
    bseg = 0;
    first = 1;
  }
  auto bound = [&](uint32_t b) -> float {
    const float* q = pl.bb + 4 * (size_t)b;
    const float x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3];
    float gx = x0 - px, gy = y0 - py;
    const float hx = px - x1, hy = py - y1;
    gx = gx > hx ? gx : hx;
    gy = gy > hy ? gy : hy;
    float m = __builtin_fabsf(px);
    const float apy = __builtin_fabsf(py), ax0 = __builtin_fabsf(x0), ay0 = __builtin_fabsf(y0), ax1 = __builtin_fabsf(x1), ay1 = __builtin_fabsf(y1);
    m = m > apy ? m : apy;
    m = m > ax0 ? m : ax0;
    m = m > ay0 ? m : ay0;
    m = m > ax1 ? m : ax1;
    m = m > ay1 ? m : ay1;
    const float E = m * 9.6e-7f;
    gx -= E;
    gy -= E;
    gx = gx > 0.0f ? gx : 0.0f;  // (NaN -> 0: never prunes)
    gy = gy > 0.0f ? gy : 0.0f;
    return (gx * gx + gy * gy) * 0.999999f;
  };
  auto scan = [&](uint32_t i0, uint32_t i1) {
    for (uint32_t i = i0; i < i1; i++) {
      float cx, cy;
      const float d = seg_closest(px, py, pl.v[i].x, pl.v[i].y, pl.v[i + 1].x, pl.v[i + 1].y, cx, cy);
      if (d < best || (d == best && i < bseg && bseg != 0xffffffffu)) {  // (an infinite distance never becomes the best)
        best = d;
        bx = cx;
        by = cy;
        bseg = i;
      }
    }
  };
  if (first < s1) {
    const uint32_t b0 = first / EG3D_BB_SEGS, b1 = (s1 - 1) / EG3D_BB_SEGS;
    // (2) the nearest block first
    uint32_t nb = b0;
    float nbound = bound(b0);
    for (uint32_t b = b0 + 1; b <= b1; b++) {
      const float t = bound(b);
      if (t < nbound) {
        nbound = t;
        nb = b;
      }
    }
    {
      const uint32_t i0 = nb * EG3D_BB_SEGS > first ? nb * EG3D_BB_SEGS : first;
      const uint32_t i1 = (nb + 1) * EG3D_BB_SEGS < s1 ? (nb + 1) * EG3D_BB_SEGS : s1;
      scan(i0, i1);
    }
    // (3) the others, ascending
    for (uint32_t b = b0; b <= b1; b++) {
      if (b == nb) continue;
      if (bound(b) > best) continue;
      const uint32_t i0 = b * EG3D_BB_SEGS > first ? b * EG3D_BB_SEGS : first;
      const uint32_t i1 = (b + 1) * EG3D_BB_SEGS < s1 ? (b + 1) * EG3D_BB_SEGS : s1;
      scan(i0, i1);
    }
  }
  out.seg = bseg;
  out.x = bx;
  out.y = by;
  return best;
}

// ---------------------------------------------------------------- grid cells ---
#if defined(__HIP_DEVICE_COMPILE__)
#define EG3D_CEILF(x) __builtin_ceilf(x)
#define EG3D_FLOORF(x) __builtin_floorf(x)
#define EG3D_FABSF(x) __builtin_fabsf(x)
#else
#define EG3D_CEILF(x) ceilf(x)
#define EG3D_FLOORF(x) floorf(x)
#define EG3D_FABSF(x) fabsf(x)
#endif

// edge_graph_3d_utilities.cpp:600-629: floor, or ceil when within 1e-3 below an integer
EG3D_HD float cell_round(float v) {
  float c = EG3D_CEILF(v);
  if ((double)(c - v) < 0.001) return c;
  return EG3D_FLOORF(v);
}
EG3D_HD bool on_cell_boundary(float m, float n) {
  float div = m / n;
  float mul = cell_round(div) * n;
  return (double)EG3D_FABSF(m - mul) < 0.001;
}
struct CellCoord {
  int32_t col, row;
  bool bx, by;  // x (resp. y) is a multiple of the cell size
};
EG3D_HD CellCoord cell_of(float cell_dim, float x, float y) {
  CellCoord c;
  c.bx = on_cell_boundary(x, cell_dim);
  c.by = on_cell_boundary(y, cell_dim);
  c.col = (int32_t)cell_round(x / cell_dim);
  c.row = (int32_t)cell_round(y / cell_dim);
  return c;
}

// Window of grid cells to visit around a point (polyLine_2d_map_search.cpp:46-77, Q7):
// empty for points on/outside the image border; shrunk on cell boundaries.
struct CellWindow {
  int32_t c0, c1, r0, r1;  // inclusive; empty if c1 < c0
};
EG3D_HD CellWindow cell_window(float cell_dim, int img_w, int img_h, int map_w, int map_h, float x, float y) {
  CellWindow w;
  w.c0 = 0;
  w.c1 = -1;
  w.r0 = 0;
  w.r1 = -1;
  if (x <= 0.0f || x >= (float)img_w || y <= 0.0f || y >= (float)img_h) return w;
  CellCoord cc = cell_of(cell_dim, x, y);
  int32_t cx = cc.col, cy = cc.row;
  if (cx < 0 || cx >= map_w) cx = map_w - 1;  // the reference compares as unsigned
  if (cy < 0 || cy >= map_h) cy = map_h - 1;
  w.r0 = cy + (cy > 0 ? -1 : 0);
  w.c0 = cx + (cx > 0 ? -1 : 0);
  // the x-boundary flag limits rows, the y-boundary flag limits columns (reference naming)
  w.r1 = cc.bx ? cy : cy + (cy < map_h - 1 ? 1 : 0);
  w.c1 = cc.by ? cx : cx + (cx < map_w - 1 ? 1 : 0);
  return w;
}

}  // namespace eg3d
