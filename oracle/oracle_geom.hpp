// ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED (the reference ships no tests or
// golden vectors and cannot be built here: OpenCV/CGAL/Boost absent — SURVEY.md F3/F4).
// CPU restatement of the 2-D geometry and polyline primitives of abignoli/EdgeGraph3D.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// Every function cites the reference lines it follows (paths relative to the reference
// tree; geometric_utilities.cpp = src/edgegraph3d/utils/geometry/geometric_utilities.cpp,
// polyline_graph_2d.cpp = src/edgegraph3d/plgs/polyline_graph_2d.cpp,
// edge_graph_3d_utilities.cpp = src/edgegraph3d/utils/edge_graph_3d_utilities.cpp).
//
// Arithmetic contract (shared with the HIP path, see DESIGN.md "Arithmetic contract"):
// no FMA contraction, no reassociation, float/double mix exactly as the reference
// compiles on x86-64 without -march flags (CMakeLists.txt:48).
// Round 4: compute_anglecos, minimum_distancesq (and squared_2d_distance inside it) are PINNED bit for bit against the
// reference's own vendored glm (tests/test_glm_pin.py, tests/glm/glm_driver.cpp, tests/golden/glm_pin_v1.npz); everything
// that does not evaluate through glm — and all OpenCV arithmetic — stays unpinned.
#pragma once
#include <cmath>
#include <cstdint>
#include <set>
#include <utility>
#include <vector>

namespace orc {

typedef unsigned long ulong_t;

struct vec2 {
  float x, y;
  vec2() : x(0), y(0) {}
  vec2(float x_, float y_) : x(x_), y(y_) {}
  bool operator==(const vec2& o) const { return x == o.x && y == o.y; }
  bool operator!=(const vec2& o) const { return !(*this == o); }
};
struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};

// geometric_utilities.cpp:555-557 — pow(float,2) promotes to double; the sum is rounded
// once to float on return (Q5).
static inline float squared_2d_distance(const vec2& a, const vec2& b) {
  double dx = (double)(a.x - b.x);
  double dy = (double)(a.y - b.y);
  return (float)(dx * dx + dy * dy);
}
// geometric_utilities.cpp:571-573
static inline float compute_2d_distance(const vec2& a, const vec2& b) {
  return std::sqrt(squared_2d_distance(a, b));
}
// glm::dot for vec2 (external/glm/glm/detail/func_geometric.inl): tmp = a*b; tmp.x+tmp.y
static inline float dot2(float ax, float ay, float bx, float by) {
  float t0 = ax * bx;
  float t1 = ay * by;
  return t0 + t1;
}
// geometric_utilities.cpp:1370-1372
static inline vec2 first_plus_ratio_of_segment(const vec2& a, const vec2& b, const float ratio) {
  float dx = b.x - a.x, dy = b.y - a.y;
  float rx = ratio * dx, ry = ratio * dy;
  return vec2(a.x + rx, a.y + ry);
}

// geometric_utilities.cpp:940-954
static inline float minimum_distancesq(const vec2& p, const vec2& v, const vec2& w, vec2& projection) {
  const float l2 = squared_2d_distance(v, w);
  if (l2 == 0.0) {
    projection = v;
    return squared_2d_distance(p, v);
  }
  const float q = dot2(p.x - v.x, p.y - v.y, w.x - v.x, w.y - v.y) / l2;
  // max<float>(0, min<float>(1, q)) with std::min/std::max semantics (NaN -> 1)
  const float m = (q < 1.0f) ? q : 1.0f;
  const float t = (0.0f < m) ? m : 0.0f;
  float ex = w.x - v.x, ey = w.y - v.y;
  float tx = t * ex, ty = t * ey;
  projection = vec2(v.x + tx, v.y + ty);
  return squared_2d_distance(p, projection);
}

// geometric_utilities.cpp:272-312. segm = (x1,y1,x2,y2), line = (a,b,c).
static inline void intersect_segment_line(float x1, float y1, float x2, float y2, const float line[3],
                                          bool& parallel, bool& overlapped, bool& intersection_found,
                                          vec2& intersection) {
  float num, den, t;
  float dx = x2 - x1, dy = y2 - y1;
  parallel = false;
  overlapped = false;
  intersection_found = false;
  {
    float t0 = line[0] * x1, t1 = line[1] * y1;
    num = (t0 + t1) + line[2];
  }
  {
    float t0 = line[0] * dx, t1 = line[1] * dy;
    den = t0 + t1;
  }
  if (den != 0) {
    t = -num / den;
    if (t >= 0 && t <= 1) {
      float tx = t * dx, ty = t * dy;
      intersection.x = x1 + tx;
      intersection.y = y1 + ty;
      intersection_found = true;
    }
  } else {
    parallel = true;
    overlapped = num == 0;
  }
}

// geometric_utilities.cpp:997-1009 (distance_point_line_sq + distance_point_line)
static inline float distance_point_line(float px, float py, const float line[3]) {
  float t0 = line[0] * px, t1 = line[1] * py;
  float den = (t0 + t1) + line[2];
  den *= den;
  float a2 = line[0] * line[0], b2 = line[1] * line[1];
  return std::sqrt(den / (a2 + b2));
}

// geometric_utilities.cpp:590-618 (get_2d_direction x2, compute_anglecos_vec2_vec2, compute_anglecos) — Q14
static inline float compute_anglecos(float x1, float y1, float x2, float y2, const float line[3]) {
  float ax = x2 - x1, ay = y2 - y1;
  float bx, by;
  if (line[1] == 0) {
    bx = 0.0f;
    by = 1.0f;
  } else {
    bx = 1.0f;
    by = -line[0] / line[1];
  }
  float d = dot2(ax, ay, bx, by);
  float aa = dot2(ax, ay, ax, ay);
  float bb = dot2(bx, by, bx, by);
  return d / std::sqrt(aa * bb);
}

// geometric_utilities.cpp:365-430
static inline void intersect_segment_line_no_quasiparallel(float x1, float y1, float x2, float y2,
                                                           const float line[3],
                                                           const float max_quasiparallel_angle_cos,
                                                           const float max_quasiparallel_dist, bool& parallel,
                                                           bool& overlapped, bool& intersection_found,
                                                           bool& quasiparallel_within_distance, bool& valid,
                                                           float& distance, vec2& intersection) {
  float num, den, t = 0;
  float dx = x2 - x1, dy = y2 - y1;
  valid = true;
  parallel = false;
  overlapped = false;
  intersection_found = false;
  quasiparallel_within_distance = false;
  {
    float t0 = line[0] * x1, t1 = line[1] * y1;
    num = (t0 + t1) + line[2];
  }
  {
    float t0 = line[0] * dx, t1 = line[1] * dy;
    den = t0 + t1;
  }
  if (den != 0) {
    t = -num / den;
    if (t >= 0 && t <= 1) {
      float tx = t * dx, ty = t * dy;
      intersection.x = x1 + tx;
      intersection.y = y1 + ty;
      distance = 0;
      intersection_found = true;
    }
    if (compute_anglecos(x1, y1, x2, y2, line) > max_quasiparallel_angle_cos) {
      if (t < 0) {
        distance = distance_point_line(x1, y1, line);
      } else if (t > 1) {
        distance = distance_point_line(x2, y2, line);
      } else {
        distance = 0;
      }
      if (distance <= max_quasiparallel_dist) {
        quasiparallel_within_distance = true;
        valid = false;
      }
    }
  } else {
    parallel = true;
    overlapped = num == 0;
    distance = distance_point_line(x1, y1, line);
    if (distance <= max_quasiparallel_dist) {
      quasiparallel_within_distance = true;
      valid = false;
    }
  }
}

// edge_graph_3d_utilities.cpp:600-629
static const double SMALL_FLOAT_EPSILON = 0.001;
static inline float floor_or_upper_if_close(const float v) {
  if ((double)(std::ceil(v) - v) < SMALL_FLOAT_EPSILON)
    return std::ceil(v);
  else
    return std::floor(v);
}
static inline bool is_m_multiple_of_n_float(const float m, const float n) {
  float div = m / n;
  float mul = floor_or_upper_if_close(div) * n;
  return (double)std::fabs(m - mul) < SMALL_FLOAT_EPSILON;
}
// returns (col,row) = (x cell, y cell); the (int) cast then widening to ulong follows :617,:624
static inline std::pair<ulong_t, ulong_t> get_2dmap_cell_from_coords(const float cell_dim, const vec2& c,
                                                                     bool& on_boundary) {
  on_boundary = is_m_multiple_of_n_float(c.x, cell_dim) || is_m_multiple_of_n_float(c.y, cell_dim);
  return std::make_pair((ulong_t)(long)(int)floor_or_upper_if_close(c.x / cell_dim),
                        (ulong_t)(long)(int)floor_or_upper_if_close(c.y / cell_dim));
}
static inline std::pair<ulong_t, ulong_t> get_2dmap_cell_from_coords(const float cell_dim, const vec2& c,
                                                                     bool& on_boundary_row,
                                                                     bool& on_boundary_col) {
  on_boundary_row = is_m_multiple_of_n_float(c.x, cell_dim);
  on_boundary_col = is_m_multiple_of_n_float(c.y, cell_dim);
  return std::make_pair((ulong_t)(long)(int)floor_or_upper_if_close(c.x / cell_dim),
                        (ulong_t)(long)(int)floor_or_upper_if_close(c.y / cell_dim));
}

// ------------------------------------------------------------------ polylines ----
// polyline_graph_2d.hpp:85-119 (pl_point), :278-294 (plg_point)
struct pl_point {
  ulong_t segment_index;
  vec2 coords;
  pl_point() : segment_index(0) {}
  pl_point(ulong_t s, const vec2& c) : segment_index(s), coords(c) {}
};
struct plg_point {
  ulong_t polyline_id;
  pl_point plp;
  plg_point() : polyline_id(0) {}
  plg_point(ulong_t id, ulong_t seg, const vec2& c) : polyline_id(id), plp(seg, c) {}
  plg_point(ulong_t id, const pl_point& p) : polyline_id(id), plp(p) {}
};

// quasi-parallel guard constants: polyline_graph_2d.hpp:73-74
static const float QP_COS = (float)0.965;
static const float QP_DIST = (float)5;
#define ORC_PL_CELL_SPLIT_RATIO (1.414 + 0.1) /* polyline_graph_2d.cpp:798-799 */

struct polyline {
  ulong_t start, end;
  std::vector<vec2> polyline_coords;
  bool valid;  // PolyLineGraph2D::is_valid_polyline, supplied by the loader
  mutable uint32_t* dir_mismatch_counter;  // Q15 instrumentation

  polyline() : start(0), end(0), valid(false), dir_mismatch_counter(nullptr) {}

  // polyline_graph_2d.cpp:901-908. With start==end (loop, Q8) this returns end (== start).
  ulong_t get_other_end(ulong_t extreme) const {
    if (extreme == start) return end;
    return start;
  }
  pl_point get_start_plp() const { return pl_point(0, polyline_coords[0]); }  // :135-137
  pl_point get_end_plp() const {                                              // :138-140
    return pl_point(polyline_coords.size() - 2, polyline_coords[polyline_coords.size() - 1]);
  }

  // polyline_graph_2d.cpp:845-862 — strict '<' keeps the first minimal segment (Q10)
  float compute_distancesq(const vec2& p, ulong_t& closest_segm, vec2& projection) const {
    float min_dist = minimum_distancesq(p, polyline_coords[0], polyline_coords[1], projection);
    closest_segm = 0;
    float cur_dist;
    vec2 cur_projection;
    for (ulong_t i = 2; i < polyline_coords.size(); i++) {
      cur_dist = minimum_distancesq(p, polyline_coords[i - 1], polyline_coords[i], cur_projection);
      if (cur_dist < min_dist) {
        min_dist = cur_dist;
        projection = cur_projection;
        closest_segm = i - 1;
      }
    }
    return min_dist;
  }

  // polyline_graph_2d.cpp:312-327 — segment passed as (v[i], v[i-1]), tagged i-1 (Q10)
  std::vector<pl_point> intersect_line(const float line[3]) const {
    std::vector<pl_point> res;
    bool parallel, overlapped, found;
    vec2 inter;
    for (ulong_t i = 1; i < polyline_coords.size(); i++) {
      intersect_segment_line(polyline_coords[i].x, polyline_coords[i].y, polyline_coords[i - 1].x,
                             polyline_coords[i - 1].y, line, parallel, overlapped, found, inter);
      if (found) res.push_back(pl_point(i - 1, inter));
    }
    return res;
  }

  // polyline_graph_2d.cpp:391-447. A direction that is neither `start` nor `end` is
  // undefined behaviour in the reference (no return statement, :445-446); the oracle
  // adopts "the walk fails" (reports the extreme as reached) and counts it — Q15.
  pl_point next_pl_point_by_distance(const pl_point init_plp, const ulong_t direction, const float distance,
                                     bool& reached_polyline_extreme) const {
    float prevdist = 0, curdist;
    reached_polyline_extreme = false;
    float ratio;
    ulong_t i;
    const ulong_t n = polyline_coords.size();
    if (direction == start) {
      curdist = compute_2d_distance(polyline_coords[init_plp.segment_index], init_plp.coords);
      if (curdist >= distance) {
        ratio = distance / curdist;
        return pl_point(init_plp.segment_index,
                        first_plus_ratio_of_segment(init_plp.coords, polyline_coords[init_plp.segment_index], ratio));
      }
      for (i = init_plp.segment_index; i > 0; i--) {
        prevdist = curdist;
        curdist = compute_2d_distance(polyline_coords[i - 1], init_plp.coords);
        if (curdist >= distance) break;
      }
      if (i == 0) {
        reached_polyline_extreme = true;
        return pl_point(0, polyline_coords[0]);
      } else {
        ratio = (distance - prevdist) / (curdist - prevdist);
        return pl_point(i - 1, first_plus_ratio_of_segment(polyline_coords[i], polyline_coords[i - 1], ratio));
      }
    } else if (direction == end) {
      if (init_plp.segment_index >= (n - 1)) {
        reached_polyline_extreme = true;
        return pl_point(n - 2, polyline_coords[n - 1]);
      }
      curdist = compute_2d_distance(polyline_coords[init_plp.segment_index + 1], init_plp.coords);
      if (curdist >= distance) {
        ratio = distance / curdist;
        return pl_point(init_plp.segment_index, first_plus_ratio_of_segment(
                                                    init_plp.coords, polyline_coords[init_plp.segment_index + 1], ratio));
      }
      for (i = init_plp.segment_index + 1; i < n - 1; i++) {
        prevdist = curdist;
        curdist = compute_2d_distance(polyline_coords[i + 1], init_plp.coords);
        if (curdist >= distance) break;
      }
      if (i == n - 1) {
        reached_polyline_extreme = true;
        return pl_point(n - 2, polyline_coords[n - 1]);
      } else {
        ratio = (distance - prevdist) / (curdist - prevdist);
        return pl_point(i, first_plus_ratio_of_segment(polyline_coords[i], polyline_coords[i + 1], ratio));
      }
    }
    if (dir_mismatch_counter) (*dir_mismatch_counter)++;
    reached_polyline_extreme = true;
    return init_plp;
  }

  // polyline_graph_2d.cpp:555-566
  std::vector<pl_point> next_pl_points_by_distance(const pl_point init_plp, const ulong_t direction,
                                                   const float distance) const {
    std::vector<pl_point> res;
    pl_point next = init_plp;
    bool reached = false;
    while (!reached) {
      next = next_pl_point_by_distance(next, direction, distance, reached);
      res.push_back(next);
    }
    return res;
  }
  // polyline_graph_2d.cpp:568-577
  std::vector<pl_point> split_equal_size_intervals(const ulong_t starting_extreme, const float distance) const {
    ulong_t direction = get_other_end(starting_extreme);
    pl_point starting_plp = (starting_extreme == start) ? get_start_plp() : get_end_plp();
    std::vector<pl_point> t_res = next_pl_points_by_distance(starting_plp, direction, distance);
    std::vector<pl_point> res;
    res.push_back(starting_plp);
    for (auto& p : t_res) res.push_back(p);
    return res;
  }
  // polyline_graph_2d.cpp:819-835 (Q7: on-boundary samples dropped)
  std::set<std::pair<ulong_t, ulong_t>> get_intersectedcells_2dmap_set(const float cell_dim) const {
    std::set<std::pair<ulong_t, ulong_t>> cells;
    std::vector<pl_point> plps = split_equal_size_intervals(start, (float)(cell_dim / ORC_PL_CELL_SPLIT_RATIO));
    std::pair<ulong_t, ulong_t> prev_cell, cur_cell;
    bool on_boundary;
    for (const auto& plp : plps) {
      cur_cell = get_2dmap_cell_from_coords(cell_dim, plp.coords, on_boundary);
      if (!on_boundary && (cells.size() == 0 || prev_cell != cur_cell)) {
        cells.insert(cur_cell);
        prev_cell = cur_cell;
      }
    }
    return cells;
  }

  // polyline_graph_2d.cpp:579-655 (bounded=false) and :657-780 (bounded=true)
  void next_pl_point_by_line_intersection_impl(const pl_point init_plp, const ulong_t direction,
                                               const float line[3], bool bounded, const float min_dist,
                                               const float max_dist, pl_point& next,
                                               bool& found_quasiparallel_segment,
                                               pl_point& next_before_quasiparallel_segment,
                                               bool& reached_polyline_extreme, bool& bounded_distance_violated,
                                               bool& found) const {
    ulong_t i;
    vec2 intersection;
    bool parallel, overlapped, intersection_found, qp, valid;
    float distance;
    const ulong_t n = polyline_coords.size();
    bounded_distance_violated = false;
    found_quasiparallel_segment = false;
    reached_polyline_extreme = false;
    found = false;

    auto accept = [&](ulong_t seg) {
      next = pl_point(seg, intersection);
      found = true;
      if (bounded) {
        float dsq = squared_2d_distance(intersection, init_plp.coords);
        if (dsq < (min_dist * min_dist) || dsq > (max_dist * max_dist)) {
          bounded_distance_violated = true;
          found = false;
        }
      }
    };

    if (direction == start) {
      const vec2& e = polyline_coords[init_plp.segment_index];
      intersect_segment_line_no_quasiparallel(init_plp.coords.x, init_plp.coords.y, e.x, e.y, line, QP_COS, QP_DIST,
                                              parallel, overlapped, intersection_found, qp, valid, distance,
                                              intersection);
      if (qp) {
        next_before_quasiparallel_segment = init_plp;
        found_quasiparallel_segment = true;
        found = false;
        return;
      } else if (intersection_found) {
        accept(init_plp.segment_index);
        return;
      }
      for (i = init_plp.segment_index; i > 0; i--) {
        const vec2& a = polyline_coords[i];
        const vec2& b = polyline_coords[i - 1];
        intersect_segment_line_no_quasiparallel(a.x, a.y, b.x, b.y, line, QP_COS, QP_DIST, parallel, overlapped,
                                                intersection_found, qp, valid, distance, intersection);
        if (qp) {
          next_before_quasiparallel_segment = pl_point(i - 1, polyline_coords[i]);
          found_quasiparallel_segment = true;
          found = false;
          return;
        } else if (intersection_found) {
          accept(i - 1);
          return;
        }
      }
      reached_polyline_extreme = true;
      found = false;
      return;
    } else if (direction == end) {
      const vec2& e = polyline_coords[init_plp.segment_index + 1];
      intersect_segment_line_no_quasiparallel(init_plp.coords.x, init_plp.coords.y, e.x, e.y, line, QP_COS, QP_DIST,
                                              parallel, overlapped, intersection_found, qp, valid, distance,
                                              intersection);
      if (qp) {
        next_before_quasiparallel_segment = init_plp;
        found_quasiparallel_segment = true;
        found = false;
        return;
      } else if (intersection_found) {
        accept(init_plp.segment_index);
        return;
      }
      for (i = init_plp.segment_index + 1; i < n - 1; i++) {
        const vec2& a = polyline_coords[i];
        const vec2& b = polyline_coords[i + 1];
        intersect_segment_line_no_quasiparallel(a.x, a.y, b.x, b.y, line, QP_COS, QP_DIST, parallel, overlapped,
                                                intersection_found, qp, valid, distance, intersection);
        if (qp) {
          next_before_quasiparallel_segment = pl_point(i, polyline_coords[i]);
          found_quasiparallel_segment = true;
          found = false;
          return;
        } else if (intersection_found) {
          accept(i);
          return;
        }
      }
      reached_polyline_extreme = true;
      found = false;
      return;
    }
    // neither start nor end: the reference only constructs an exception object and
    // leaves every out flag false (polyline_graph_2d.cpp:653-654, :778-779) — Q15
    if (dir_mismatch_counter) (*dir_mismatch_counter)++;
  }
  void next_pl_point_by_line_intersection(const pl_point init_plp, const ulong_t direction, const float line[3],
                                          pl_point& next, bool& found_qp, pl_point& next_before_qp,
                                          bool& reached_extreme, bool& found) const {
    bool bdv;
    next_pl_point_by_line_intersection_impl(init_plp, direction, line, false, 0, 0, next, found_qp, next_before_qp,
                                            reached_extreme, bdv, found);
  }
  void next_pl_point_by_line_intersection_bounded_distance(const pl_point init_plp, const ulong_t direction,
                                                           const float line[3], const float min_dist,
                                                           const float max_dist, pl_point& next, bool& found_qp,
                                                           pl_point& next_before_qp, bool& reached_extreme,
                                                           bool& bounded_distance_violated, bool& found) const {
    next_pl_point_by_line_intersection_impl(init_plp, direction, line, true, min_dist, max_dist, next, found_qp,
                                            next_before_qp, reached_extreme, bounded_distance_violated, found);
  }
};

// One view's polyline graph as far as the path reads it (polyline_graph_2d.hpp:222).
struct PLG {
  std::vector<polyline> polylines;
};

// ---------------------------------------------------------------- grid maps ----
// PolyLine2DMap + PolyLine2DMapSearch: polyLine_2d_map.cpp:40-58, polyLine_2d_map_search.cpp:43-88
struct PolyLine2DMapSearch {
  const PLG* plg;
  int img_w, img_h;
  float cell_dim;
  int map_w, map_h;                           // mapsz
  std::vector<std::vector<ulong_t>> cells;    // [row*map_w + col] == pls_id_maps[row][col]
  uint32_t dropped_out_of_range;              // guard for the reference's unchecked index (:57)

  PolyLine2DMapSearch() : plg(nullptr), img_w(0), img_h(0), cell_dim(0), map_w(0), map_h(0), dropped_out_of_range(0) {}
  void build(const PLG& g, int w, int h, float search_dist) {
    plg = &g;
    img_w = w;
    img_h = h;
    cell_dim = search_dist;
    map_w = int(std::ceil(img_w / cell_dim));
    map_h = int(std::ceil(img_h / cell_dim));
    cells.assign((size_t)map_w * map_h, std::vector<ulong_t>());
    dropped_out_of_range = 0;
    for (ulong_t pl_id = 0; pl_id < g.polylines.size(); pl_id++) {
      if (!g.polylines[pl_id].valid) continue;  // add_polyline :52-53
      auto cs = g.polylines[pl_id].get_intersectedcells_2dmap_set(cell_dim);
      for (const auto& c : cs) {
        if (c.first >= (ulong_t)map_w || c.second >= (ulong_t)map_h) {
          dropped_out_of_range++;
          continue;
        }
        cells[c.second * map_w + c.first].push_back(pl_id);  // [row=second][col=first] (Q7)
      }
    }
  }
  // polyLine_2d_map_search.cpp:46-77
  std::set<ulong_t> find_polylines_potentially_within_search_dist(const vec2& coords) const {
    std::set<ulong_t> res;
    bool on_boundary_row, on_boundary_col;
    if (coords.x <= 0 || coords.x >= img_w || coords.y <= 0 || coords.y >= img_h) return res;
    std::pair<ulong_t, ulong_t> cc = get_2dmap_cell_from_coords(cell_dim, coords, on_boundary_row, on_boundary_col);
    long cx = cc.first >= (ulong_t)map_w ? map_w - 1 : (long)cc.first;
    long cy = cc.second >= (ulong_t)map_h ? map_h - 1 : (long)cc.second;
    int i0 = cy > 0 ? -1 : 0, j0 = cx > 0 ? -1 : 0;
    int i1, j1;
    // note the reference's naming: on_boundary_row is the x test and limits i (rows) — :59-74
    if (!on_boundary_row && !on_boundary_col) {
      i1 = (cy < map_h - 1 ? 1 : 0);
      j1 = (cx < map_w - 1 ? 1 : 0);
    } else if (on_boundary_row && !on_boundary_col) {
      i1 = 0;
      j1 = (cx < map_w - 1 ? 1 : 0);
    } else if (!on_boundary_row && on_boundary_col) {
      i1 = (cy < map_h - 1 ? 1 : 0);
      j1 = 0;
    } else {
      i1 = 0;
      j1 = 0;
    }
    for (int i = i0; i <= i1; i++)
      for (int j = j0; j <= j1; j++) {
        const auto& v = cells[(cy + i) * map_w + (cx + j)];
        res.insert(v.begin(), v.end());
      }
    return res;
  }
  // polyLine_2d_map_search.cpp:81-88
  void find_unique_polyline_potentially_within_search_dist(const vec2& coords, ulong_t& pl_id, bool& valid) const {
    valid = false;
    std::set<ulong_t> tmp = find_polylines_potentially_within_search_dist(coords);
    if (tmp.size() == 1) {
      valid = true;
      pl_id = *(tmp.begin());
    }
  }
};

}  // namespace orc
