// ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED.
// CPU restatement of the step before the path: binary edge image -> pixel graph ->
// PolyLineGraph2DHMapImpl -> optimize(). Kept in the reference's own shape (std::set adjacency,
// unordered_map of coordinates, vector-of-vectors connections) as the independent check of the
// product's flat-array builder (edgegraph3d_amd/host/plg_build.cpp). Follows, relative to
// /root/reference:
//   convertEdgeImagesPixelToNodesNoSquaresNoTriangles_remove_useless_hubs, convertEdgeImagePixelToGraph_NoCycles
//       src/edgegraph3d/io/input/convert_edge_images_pixel_to_segment.cpp:294-426
//   GraphAdjacencySetNoType::add_edge / is_connected(.., max_dist)   src/edgegraph3d/plgs/graph_adjacency_set_no_type.cpp:73-150
//   find_polylines & friends, convert_EdgeGraph_to_PolyLineGraph     convert_edge_images_pixel_to_segment.cpp:428-626
//   convertEdgeImagePolyLineGraph_optimized                           :868-883
//   PolyLineGraph2DHMapImpl::get_node_id / add_polyline / filter_polyline / is_duplicate / add_direct_connection /
//       connect_close_extremes / remove_2connection_nodes / remove_degenerate_loops / remove_invalid_polylines /
//       split_loop(s) / split_polyline / optimize       src/edgegraph3d/plgs/polyline_graph_2d_hmap_impl.cpp:47-266
//   polyline::update_length / compute_max_smooth_length / split / next_pl_point_by_length / intersect_segment /
//       merge_polylines / simplify_polyline & helpers / operator== ; PolyLineGraph2D::remove_connection /
//       remove_polyline / invalidate_node / is_valid_node / is_valid_polyline / optimize / is_extreme / has_loop /
//       get_extreme_nodes_ids_and_coords / find_closest_pairs_with_max_dist / compute_components /
//       compute_components_with_polylines / filter_components_by_polylinesmoothlength / intersect_polylines
//       src/edgegraph3d/plgs/polyline_graph_2d.cpp:76-98,166-182,297-310,449-497,905-1160,1315-1355,1423-1450,1478-1488,1926-2066
//   geometry: compute_2dline, distance_point_line_sq, intersect_segment_line/_segment, point_in_segment_bounding_box,
//       compute_anglecos_vec2_vec2, middle_point      src/edgegraph3d/utils/geometry/geometric_utilities.cpp:272-312,432-442,579-588,997-1001,1341-1364
//
// Where the reference's behaviour is undefined the restatement fixes a rule (the product adopts
// the same one):
//   U1 pixels read outside the image (the "useless hub" test reads (i+1,j+1), (i+1,j-1), ... without
//      bounds checks): the image is addressed as ONE row-major buffer, as cv::Mat memory is — column -1
//      is the last pixel of the previous row, column `cols` the first of the next — and anything
//      outside the buffer is "not an edge".
//   U2 GraphAdjacencySetNoType::is_connected(a, b, max_dist) declares `vector<bool> visited_vec` and
//      pushes node ids into it, so the per-call reset `visited[v] = false` only ever clears entries 0
//      and 1: the `visited` member keeps every node any earlier call has touched. That is defined
//      behaviour and is reproduced exactly (it decides which pixel-graph edges are added).
#pragma once
#include <algorithm>
#include <cstdint>
#include <limits>
#include <set>
#include <stack>
#include <unordered_map>
#include <vector>

#include "oracle_geom.hpp"

namespace orc {
namespace n2 {

static const float INVALID_COORD = -1.0f;            // INVALID_POINT_COORDS
static const float INVALID_LENGTH = -1.0f;           // INVALID_POLYLINE_LENGTH
static const float DIRECT_CONNECTION_MAXDIST = 6.0f;  // DIRECT_CONNECTION_EXTREMES_MAXDIST
static const float MAX_LINEARIZABILITY_DIST = 1.0f;   // MAXIMUM_LINEARIZABILITY_DISTANCE
static const double TOP_FILTER = 0.82;                // TOP_FILTER_BY_POLYLINESMOOTHLENGTH
static const unsigned LOOP_CHECK_DIST = 8;

// ---- geometry --------------------------------------------------------------------------------
struct line3 {
  float a, b, c;
};
static inline line3 compute_2dline(const vec2& a, const vec2& b) {
  if (a.x == b.x) return line3{1.0f, 0.0f, -a.x};
  float m = (b.y - a.y) / (b.x - a.x);
  float q = a.y - m * a.x;
  return line3{m, -1.0f, q};
}
static inline float distance_point_line_sq(const vec2& p, const line3& l) {
  float den = l.a * p.x + l.b * p.y + l.c;
  den *= den;
  return den / (l.a * l.a + l.b * l.b);
}
// intersect_segment_line (geometric_utilities.cpp:272-312), segment (s0 -> s1)
static inline void intersect_segment_line2(const vec2& s0, const vec2& s1, const line3& l, bool& found, vec2& inter) {
  const float dx = s1.x - s0.x, dy = s1.y - s0.y;
  found = false;
  const float num = l.a * s0.x + l.b * s0.y + l.c;
  const float den = l.a * dx + l.b * dy;
  if (den != 0) {
    const float t = -num / den;
    if (t >= 0 && t <= 1) {
      inter.x = s0.x + t * dx;
      inter.y = s0.y + t * dy;
      found = true;
    }
  }
}
static inline bool point_in_bbox(const vec2& a, const vec2& b, const vec2& p) {
  return ((a.x <= p.x && p.x <= b.x) || (b.x <= p.x && p.x <= a.x)) && ((a.y <= p.y && p.y <= b.y) || (b.y <= p.y && p.y <= a.y));
}
// intersect_segment_segment(segm1, segm2): line through segm1, intersected with segm2, inside segm1's box
static inline bool intersect_segment_segment(const vec2& a0, const vec2& a1, const vec2& b0, const vec2& b1, vec2& inter) {
  const line3 l = compute_2dline(a0, a1);
  bool found;
  intersect_segment_line2(b0, b1, l, found, inter);
  return found && point_in_bbox(a0, a1, inter);
}
static inline float anglecos(const vec2& a1, const vec2& a2, const vec2& b1, const vec2& b2) {
  const float ax = a2.x - a1.x, ay = a2.y - a1.y, bx = b2.x - b1.x, by = b2.y - b1.y;
  return (ax * bx + ay * by) / std::sqrt((ax * ax + ay * ay) * (bx * bx + by * by));
}

// ---- pixel graph -----------------------------------------------------------------------------
struct PixelGraph {
  std::vector<std::set<ulong_t>> adj;
  std::vector<bool> visited;  // U2: never really reset
  explicit PixelGraph(size_t n) : adj(n), visited(n, false) {}
  void add_edge(ulong_t a, ulong_t b) {
    adj[a].insert(b);
    adj[b].insert(a);
  }
  bool is_connected(ulong_t start_node, ulong_t end_node, ulong_t max_dist) {
    std::vector<bool> visited_vec;
    visited_vec.push_back(start_node);  // (sic) a bool
    visited[start_node] = true;
    bool found = false;
    std::stack<ulong_t> cur;
    cur.push(start_node);
    ulong_t cur_dist = 0;
    while (cur_dist <= max_dist && !cur.empty()) {
      std::stack<ulong_t> next;
      while (!found && !cur.empty()) {
        const ulong_t n = cur.top();
        cur.pop();
        for (const auto c : adj[n]) {
          if (c == end_node) {
            found = true;
            break;
          }
          if (!visited[c]) {
            next.push(c);
            visited[c] = true;
            visited_vec.push_back(c);  // (sic) a bool
          }
        }
      }
      for (bool v : visited_vec) {  // (sic) only entries 0 / 1 are ever cleared
        const size_t idx = v ? 1 : 0;
        if (idx < visited.size()) visited[idx] = false;
      }
      cur = next;
      cur_dist++;
    }
    return found;
  }
};

// mask: rows*cols bytes, 1 = edge colour. Modified in place (the "useless hub" pixels are cleared).
static inline void image_to_graph(uint8_t* mask, int rows, int cols, PixelGraph*& graph, std::vector<vec2>& node_coords) {
  const long total = (long)rows * cols;
  auto edge = [&](int i, int j) -> bool {  // U1
    const long idx = (long)i * cols + j;
    return idx >= 0 && idx < total && mask[idx] != 0;
  };
  std::vector<ulong_t> node_id((size_t)total, 0);
  node_coords.clear();
  ulong_t next_id = 0;
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++)
      if (edge(i, j)) {
        if ((i > 1 && j > 1 && edge(i - 1, j) && edge(i, j - 1) && !edge(i + 1, j + 1)) ||
            (i > 1 && j < cols - 1 && edge(i - 1, j) && edge(i, j + 1) && !edge(i + 1, j - 1)) ||
            (i < rows - 1 && j < cols - 1 && edge(i + 1, j) && edge(i, j + 1) && !edge(i - 1, j - 1)) ||
            (i < rows - 1 && j > 1 && edge(i + 1, j) && edge(i, j - 1) && !edge(i - 1, j + 1))) {
          mask[(size_t)i * cols + j] = 0;  // clear pixel
        } else {
          node_id[(size_t)i * cols + j] = next_id++;
          node_coords.push_back(vec2((float)(j + 0.5), (float)(i + 0.5)));
        }
      }
  graph = new PixelGraph(next_id);
  PixelGraph& g = *graph;
  auto try_edge = [&](ulong_t p, int cy, int cx) {
    if (mask[(size_t)cy * cols + cx]) {
      const ulong_t c = node_id[(size_t)cy * cols + cx];
      if (p != c && !g.is_connected(p, c, LOOP_CHECK_DIST)) g.add_edge(p, c);
    }
  };
  for (int i = 0; i < rows - 1; i++)
    for (int j = 0; j < cols - 1; j++)
      if (mask[(size_t)i * cols + j]) {
        const ulong_t p = node_id[(size_t)i * cols + j];
        try_edge(p, i, j + 1);
        try_edge(p, i + 1, j);
        try_edge(p, i + 1, j + 1);
        if (j > 1) try_edge(p, i + 1, j - 1);
      }
}

// ---- graph -> polylines ------------------------------------------------------------------------
static inline ulong_t neighbour_no_come_back(const std::set<ulong_t>& a, ulong_t no_come_back) {
  auto it = a.begin();
  const ulong_t prev = *it;
  ++it;
  const ulong_t next = *it;
  return prev != no_come_back ? prev : next;
}
static inline void find_polylineend_no_come_back(ulong_t start, const std::vector<std::set<ulong_t>>& adj, ulong_t no_come_back,
                                                 std::vector<ulong_t>& res) {
  ulong_t prev = no_come_back, cur = start;
  res.push_back(cur);
  while (cur != no_come_back && adj[cur].size() == 2) {
    const ulong_t nx = neighbour_no_come_back(adj[cur], prev);
    prev = cur;
    cur = nx;
    res.push_back(cur);
  }
}
static inline std::vector<std::vector<ulong_t>> find_polylines(ulong_t start, const std::vector<std::set<ulong_t>>& adj) {
  std::vector<std::vector<ulong_t>> res;
  const std::set<ulong_t>& cur_adj = adj[start];
  const size_t n = cur_adj.size();
  if (n == 2) {
    std::vector<ulong_t> r;
    auto it = cur_adj.begin();
    const ulong_t prev = *it;
    ++it;
    const ulong_t next = *it;
    find_polylineend_no_come_back(prev, adj, start, r);
    std::reverse(r.begin(), r.end());
    r.push_back(start);
    if (r[0] != r[r.size() - 1]) find_polylineend_no_come_back(next, adj, start, r);
    res.push_back(r);
  } else if (n == 1) {
    std::vector<ulong_t> r;
    r.push_back(start);
    find_polylineend_no_come_back(*cur_adj.begin(), adj, start, r);
    res.push_back(r);
  } else if (n > 2) {
    for (auto it = cur_adj.begin(); it != cur_adj.end(); ++it) {
      std::vector<ulong_t> r;
      r.push_back(start);
      find_polylineend_no_come_back(*it, adj, start, r);
      res.push_back(r);
    }
  }
  return res;
}

// ---- the polyline graph ------------------------------------------------------------------------
struct KeyFuncs2 {
  size_t operator()(const vec2& k) const { return std::hash<int>()((int)k.x) ^ std::hash<int>()((int)k.y); }
  bool operator()(const vec2& a, const vec2& b) const { return a.x == b.x && a.y == b.y; }
};

struct PL {
  ulong_t start, end;
  std::vector<vec2> coords;
  float length;
  void update_length() {
    length = 0.0f;
    for (size_t i = 1; i < coords.size(); i++) length += compute_2d_distance(coords[i], coords[i - 1]);
  }
  bool is_loop() const { return start == end; }
  ulong_t other_end(ulong_t e) const { return e == start ? end : start; }  // (neither: UB in the reference; unused so)
  float max_smooth_length() const {
    float maxl = 0.0f;
    size_t i = 1;
    while (i < coords.size()) {
      float cur = compute_2d_distance(coords[i], coords[i - 1]);
      for (i++; i < coords.size(); i++)
        if (anglecos(coords[i - 1], coords[i], coords[i - 2], coords[i - 1]))  // (sic) truthiness of the cosine
          cur += compute_2d_distance(coords[i], coords[i - 1]);
        else
          break;
      maxl = maxl < cur ? cur : maxl;
    }
    return maxl;
  }
};

static inline bool linearizable(const std::vector<vec2>& c, size_t start, size_t end, float maxsq) {
  const line3 l = compute_2dline(c[start], c[end]);
  for (size_t i = start + 1; i < end; i++)
    if (distance_point_line_sq(c[i], l) > maxsq) return false;
  return true;
}
static inline size_t find_max_se(const std::vector<vec2>& c, size_t start, size_t max_se, float maxsq) {
  if (max_se <= start) return start;
  for (size_t se = max_se; se > start + 1; se--)
    if (linearizable(c, start, se, maxsq)) return se;
  return start + 1;
}
static inline size_t find_min_eb(const std::vector<vec2>& c, size_t end, size_t min_eb, float maxsq) {
  if (min_eb >= end) return end;
  for (size_t eb = min_eb; eb < end - 1; eb++)
    if (linearizable(c, eb, end, maxsq)) return eb;
  return end - 1;
}
static inline std::vector<vec2> simplify_polyline(const std::vector<vec2>& c, float max_dist) {
  const float maxsq = max_dist * max_dist;
  size_t start = 0, end = c.size() - 1;
  std::vector<vec2> head, tail;
  head.push_back(c[start]);
  tail.push_back(c[end]);
  while (end > start + 1) {
    size_t se, eb;
    {  // find_compatible_se_eb
      size_t max_se = end, min_eb = start;
      do {
        se = find_max_se(c, start, max_se, maxsq);
        if (se == end) {
          eb = 0;
          break;
        }
        eb = find_min_eb(c, end, min_eb, maxsq);
        max_se--;
        min_eb++;
      } while (eb < se);
    }
    if (se == end) break;
    if (se == eb) {
      head.push_back(c[se]);
    } else {
      head.push_back(c[se]);
      tail.push_back(c[eb]);
    }
    start = se;
    end = eb;
  }
  for (auto it = tail.rbegin(); it != tail.rend(); ++it) head.push_back(*it);
  return head;
}

struct PLG2 {
  std::vector<PL> polylines;
  std::vector<std::vector<ulong_t>> connections;
  std::vector<vec2> nodes_coords;
  std::unordered_map<vec2, ulong_t, KeyFuncs2, KeyFuncs2> point_map;

  bool is_valid_node(ulong_t n) const { return nodes_coords[n].x != INVALID_COORD && nodes_coords[n].y != INVALID_COORD; }
  bool is_valid_polyline(ulong_t p) const {
    const PL& pl = polylines[p];
    return is_valid_node(pl.start) && is_valid_node(pl.end) && pl.coords.size() > 1 && nodes_coords[pl.start] == pl.coords[0] &&
           nodes_coords[pl.end] == pl.coords[pl.coords.size() - 1];
  }
  // PolyLineGraph2D::invalidate_node (the base one: no point_map erase)
  void base_invalidate_node(ulong_t n) {
    nodes_coords[n] = vec2(INVALID_COORD, INVALID_COORD);
    // for pid : connections[n] remove_polyline(pid) — only ever reached with an empty list (see header)
    std::vector<ulong_t> snapshot = connections[n];
    for (auto pid : snapshot) remove_polyline(pid);
    connections[n].clear();
  }
  void hmap_invalidate_node(ulong_t n) {
    point_map.erase(nodes_coords[n]);
    base_invalidate_node(n);
  }
  void remove_connection(ulong_t n, ulong_t pid) {
    auto& c = connections[n];
    c.erase(std::remove(c.begin(), c.end(), pid), c.end());
    if (c.size() == 0) base_invalidate_node(n);
  }
  void remove_polyline(ulong_t pid) {
    PL& p = polylines[pid];
    remove_connection(p.start, pid);
    remove_connection(p.end, pid);
    p.coords.clear();
    p.update_length();
    p.length = INVALID_LENGTH;
  }
  ulong_t get_node_id(const vec2& c) {
    auto it = point_map.find(c);
    if (it != point_map.end() && !is_valid_node(it->second)) {
      hmap_invalidate_node(it->second);
      it = point_map.end();
    }
    if (it == point_map.end()) {
      const ulong_t id = nodes_coords.size();
      point_map[c] = id;
      connections.push_back(std::vector<ulong_t>());
      nodes_coords.push_back(c);
      return id;
    }
    return it->second;
  }
  static bool vec_eq(const std::vector<vec2>& a, const std::vector<vec2>& b, bool inv) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
      if (!(a[i] == (inv ? b[a.size() - i - 1] : b[i]))) return false;
    return true;
  }
  bool pl_equal(const PL& a, const PL& b) const {
    return (a.start == b.start && a.end == b.end && vec_eq(a.coords, b.coords, false)) ||
           (a.start == b.end && a.end == b.start && vec_eq(a.coords, b.coords, true));
  }
  bool is_duplicate(const PL& pl) const {
    const auto& s = connections[pl.start];
    const auto& e = connections[pl.end];
    const auto& smallest = s.size() < e.size() ? s : e;
    for (auto id : smallest)
      if (pl_equal(polylines[id], pl)) return true;
    return false;
  }
  void internal_add_polyline(const PL& pl) {
    if (!is_duplicate(pl)) {
      const ulong_t id = polylines.size();
      polylines.push_back(pl);
      connections[pl.start].push_back(id);
      if (pl.start != pl.end) connections[pl.end].push_back(id);
    }
  }
  void add_polyline(const std::vector<vec2>& in) {
    std::vector<vec2> c;
    if (in[0] == in[in.size() - 1] && in.size() == 4 && squared_2d_distance(in[1], in[2]) <= 4) {  // filter_polyline
      c.push_back(in[0]);
      c.push_back(vec2((in[1].x + in[2].x) / 2, (in[1].y + in[2].y) / 2));
    } else
      c = in;
    PL pl;
    pl.start = get_node_id(c[0]);
    pl.end = get_node_id(c[c.size() - 1]);
    pl.coords = c;
    pl.update_length();
    internal_add_polyline(pl);
  }
  void add_direct_connection(ulong_t a, ulong_t b) {
    PL pl;
    pl.start = a;
    pl.end = b;
    pl.coords = {nodes_coords[a], nodes_coords[b]};
    pl.update_length();
    internal_add_polyline(pl);
  }
  bool is_extreme(ulong_t n) const { return connections[n].size() == 1 && !polylines[connections[n][0]].is_loop(); }

  // ---- optimize() stages
  void remove_invalid_polylines() {
    const size_t n = polylines.size();
    for (size_t i = 0; i < n; i++)
      if (!is_valid_polyline(i)) remove_polyline(i);
  }
  void remove_degenerate_loops() {
    const size_t n = polylines.size();
    for (size_t i = 0; i < n; i++)
      if (is_valid_polyline(i)) {
        const PL& p = polylines[i];
        if (p.start == p.end || p.coords[0] == p.coords[p.coords.size() - 1])
          if (p.coords.size() < 5) remove_polyline(i);
      }
  }
  static PL merge_polylines(const PL& p1, const PL& p2) {
    PL r;
    auto app = [&](const std::vector<vec2>& v, bool rev, bool skip_first) {
      if (!rev)
        for (size_t i = skip_first ? 1 : 0; i < v.size(); i++) r.coords.push_back(v[i]);
      else
        for (size_t k = skip_first ? 1 : 0; k < v.size(); k++) r.coords.push_back(v[v.size() - 1 - k]);
    };
    if (p1.start == p2.start) {
      r.start = p1.end;
      r.end = p2.end;
      app(p1.coords, true, false);
      app(p2.coords, false, true);
    } else if (p1.start == p2.end) {
      r.start = p2.start;
      r.end = p1.end;
      app(p2.coords, false, false);
      app(p1.coords, false, true);
    } else if (p1.end == p2.start) {
      r.start = p1.start;
      r.end = p2.end;
      app(p1.coords, false, false);
      app(p2.coords, false, true);
    } else {  // p1.end == p2.end
      r.start = p1.start;
      r.end = p2.start;
      app(p1.coords, false, false);
      app(p2.coords, true, true);
    }
    r.update_length();
    return r;
  }
  void remove_2connection_nodes() {
    for (ulong_t node = 0; node < connections.size(); node++)
      if (connections[node].size() == 2) {
        const ulong_t id1 = connections[node][0], id2 = connections[node][1];
        const ulong_t o1 = polylines[id1].other_end(node), o2 = polylines[id2].other_end(node);
        if (vec_eq(polylines[id1].coords, polylines[id2].coords, false) || vec_eq(polylines[id1].coords, polylines[id2].coords, true)) {
          remove_polyline(id2);
          continue;
        }
        if (o1 != node && o2 != node) {
          const PL p3 = merge_polylines(polylines[id1], polylines[id2]);
          internal_add_polyline(p3);
          remove_polyline(id1);
          remove_polyline(id2);
          hmap_invalidate_node(node);
        }
      }
  }
  void simplify_all() {
    const size_t n = polylines.size();
    for (size_t i = 0; i < n; i++)
      if (is_valid_polyline(i)) {
        polylines[i].coords = simplify_polyline(polylines[i].coords, MAX_LINEARIZABILITY_DIST);
        polylines[i].update_length();
      }
  }
  void compute_components(std::vector<ulong_t>& comp_of_node, std::vector<std::set<ulong_t>>& comps) const {
    const size_t N = nodes_coords.size();
    comp_of_node.assign(N, 0);
    comps.clear();
    std::vector<bool> explored(N, false), in_to_explore(N, false);
    std::stack<ulong_t> to_explore;
    ulong_t cur_id = 0;
    for (ulong_t s = 0; s < N; s++)
      if (!explored[s]) {
        explored[s] = true;
        std::set<ulong_t> cur;
        cur.insert(s);
        comp_of_node[s] = cur_id;
        for (auto p : connections[s]) {
          const ulong_t o = polylines[p].other_end(s);
          in_to_explore[o] = true;
          to_explore.push(o);
        }
        while (!to_explore.empty()) {
          const ulong_t n = to_explore.top();
          to_explore.pop();
          cur.insert(n);
          comp_of_node[n] = cur_id;
          in_to_explore[n] = false;
          explored[n] = true;
          for (auto p : connections[n]) {
            const ulong_t o = polylines[p].other_end(n);
            if (!explored[o] || o == n)
              if (!in_to_explore[o] && o != n) to_explore.push(o);
          }
        }
        comps.push_back(cur);
        cur_id++;
      }
  }
  bool any_polyline_intersects(const vec2& a, const vec2& b) const {  // intersect_polylines(segment).size() > 0
    for (size_t i = 0; i < polylines.size(); i++)
      if (is_valid_polyline(i)) {
        const auto& c = polylines[i].coords;
        for (size_t k = 1; k < c.size(); k++) {
          vec2 inter;
          if (intersect_segment_segment(c[k], c[k - 1], a, b, inter)) return true;
        }
      }
    return false;
  }
  void connect_close_extremes() {
    std::vector<ulong_t> ids;
    std::vector<vec2> pts;
    for (ulong_t n = 0; n < nodes_coords.size(); n++)
      if (is_valid_node(n) && is_extreme(n)) {
        ids.push_back(n);
        pts.push_back(nodes_coords[n]);
      }
    // find_closest_pairs_with_max_dist: reciprocal nearest neighbours within the distance
    std::vector<std::pair<ulong_t, ulong_t>> pairs;
    {
      const float maxsq = DIRECT_CONNECTION_MAXDIST * DIRECT_CONNECTION_MAXDIST;
      std::vector<ulong_t> closest(pts.size());
      for (ulong_t i = 0; i < pts.size(); i++) {
        float mind = std::numeric_limits<float>::max();
        ulong_t mid = (ulong_t)-1;
        for (ulong_t j = 0; j < pts.size(); j++)
          if (j != i) {
            const float d = squared_2d_distance(pts[i], pts[j]);
            if (d < mind) {
              mind = d;
              mid = j;
            }
          }
        closest[i] = mid;
        if (mid < i)
          if (i == closest[mid] && squared_2d_distance(pts[i], pts[mid]) <= maxsq) pairs.push_back(std::make_pair(ids[i], ids[mid]));
      }
    }
    std::vector<ulong_t> comp_of_node;
    std::vector<std::set<ulong_t>> comps;
    compute_components(comp_of_node, comps);
    for (const auto& pp : pairs)
      if (comp_of_node[pp.first] != comp_of_node[pp.second])
        if (!any_polyline_intersects(nodes_coords[pp.first], nodes_coords[pp.second])) {
          add_direct_connection(pp.first, pp.second);
          ulong_t new_id, to_change;
          if (comps[comp_of_node[pp.first]].size() < comps[comp_of_node[pp.second]].size()) {
            new_id = comp_of_node[pp.second];
            to_change = comp_of_node[pp.first];
          } else {
            new_id = comp_of_node[pp.first];
            to_change = comp_of_node[pp.second];
          }
          for (auto n : comps[to_change]) comp_of_node[n] = new_id;
        }
  }
  // next_pl_point_by_length from the START extreme towards node `direction` (the only use: split_loop)
  bool midpoint_by_length(const PL& p, ulong_t direction, float length, ulong_t& seg, vec2& out) const {
    // returns reached_polyline_extreme
    const auto& c = p.coords;
    const ulong_t init_seg = 0;
    const vec2 init = c[0];
    float prevlen = 0, curlen;
    ulong_t i;
    if (direction == p.start) {
      curlen = compute_2d_distance(c[init_seg], init);
      if (curlen >= length) {
        const float ratio = length / curlen;
        seg = init_seg;
        out = first_plus_ratio_of_segment(init, c[init_seg], ratio);
        return false;
      }
      for (i = init_seg; i > 0; i--) {
        prevlen = curlen;
        curlen += compute_2d_distance(c[i - 1], init);
        if (curlen >= length) break;
      }
      if (i == 0) {
        seg = 0;
        out = c[0];
        return true;
      }
      const float ratio = (length - prevlen) / (curlen - prevlen);
      seg = i - 1;
      out = first_plus_ratio_of_segment(c[i], c[i - 1], ratio);
      return false;
    }
    // direction == end
    curlen = compute_2d_distance(c[init_seg + 1], init);
    if (curlen >= length) {
      const float ratio = length / curlen;
      seg = init_seg;
      out = first_plus_ratio_of_segment(init, c[init_seg + 1], ratio);
      return false;
    }
    for (i = init_seg + 1; i < c.size() - 1; i++) {
      prevlen = curlen;
      curlen += compute_2d_distance(c[i + 1], init);  // (sic) distance from the initial point
      if (curlen >= length) break;
    }
    if (i == c.size() - 1) {
      seg = c.size() - 2;
      out = c[c.size() - 1];
      return true;
    }
    const float ratio = (length - prevlen) / (curlen - prevlen);
    seg = i;
    out = first_plus_ratio_of_segment(c[i], c[i + 1], ratio);
    return false;
  }
  void split_polyline(ulong_t pid, ulong_t seg, const vec2& at) {
    get_node_id(at);
    const std::vector<vec2> c = polylines[pid].coords;
    std::vector<vec2> c1, c2;
    for (ulong_t i = 0; i <= seg; i++) c1.push_back(c[i]);
    if (at != c[seg]) c1.push_back(at);
    c2.push_back(at);
    for (ulong_t i = seg + 1; i < c.size(); i++) c2.push_back(c[i]);
    remove_polyline(pid);
    add_polyline(c1);
    add_polyline(c2);
  }
  void split_loops() {
    const size_t n = polylines.size();
    for (size_t i = 0; i < n; i++)
      if (is_valid_polyline(i)) {
        const PL& p = polylines[i];
        if (p.length >= 10 && p.is_loop()) {  // MINSPLITLOOP_LENGTH
          ulong_t seg;
          vec2 mid;
          // p.get_extreme_plp(p.start) is the start point; the walk goes "towards p.end", which for a loop IS
          // p.start, so it stops at once: reached_polyline_extreme is true and nothing is split
          const bool reached = midpoint_by_length(p, p.end, p.length / 2, seg, mid);
          if (!reached) split_polyline(i, seg, mid);
        }
      }
  }
  void filter_components_by_polylinesmoothlength() {
    std::vector<ulong_t> comp_of_node;
    std::vector<std::set<ulong_t>> comps;
    compute_components(comp_of_node, comps);
    std::vector<std::set<ulong_t>> pl_comps(comps.size());
    for (ulong_t n = 0; n < nodes_coords.size(); n++)
      if (is_valid_node(n))
        for (auto pid : connections[n]) pl_comps[comp_of_node[n]].insert(pid);
    std::vector<float> smooth(polylines.size(), 0.0f);
    for (size_t i = 0; i < polylines.size(); i++)
      if (is_valid_polyline(i)) smooth[i] = polylines[i].max_smooth_length();
    std::vector<float> cpy = smooth;
    const size_t idx = (size_t)(smooth.size() * TOP_FILTER);
    if (smooth.empty()) return;  // (nth_element on an empty range: undefined in the reference)
    std::nth_element(smooth.begin(), smooth.begin() + idx, smooth.end());
    const float filter = smooth[idx];
    for (size_t c = 0; c < pl_comps.size(); c++) {
      bool remove = true;
      for (auto pid : pl_comps[c])
        if (cpy[pid] >= filter) {
          remove = false;
          break;
        }
      if (remove)
        for (auto pid : pl_comps[c]) remove_polyline(pid);
    }
  }
  void optimize() {
    remove_invalid_polylines();
    remove_degenerate_loops();
    remove_2connection_nodes();
    simplify_all();
    connect_close_extremes();
    simplify_all();
    split_loops();
    filter_components_by_polylinesmoothlength();
  }
};

// convertEdgeImagePolyLineGraph_optimized
static inline void edge_image_to_plg(uint8_t* mask, int rows, int cols, PLG2& plg) {
  PixelGraph* g = nullptr;
  std::vector<vec2> nodes;
  image_to_graph(mask, rows, cols, g, nodes);
  const std::vector<std::set<ulong_t>>& adj = g->adj;
  std::vector<bool> processed(nodes.size(), false);
  for (ulong_t i = 0; i < nodes.size(); i++)
    if (!processed[i]) {
      const auto pls = find_polylines(i, adj);
      for (const auto& ids : pls) {
        const ulong_t cs = ids[0], ce = ids[ids.size() - 1];
        plg.get_node_id(nodes[cs]);
        plg.get_node_id(nodes[ce]);
        if (!(adj[cs].size() > 2)) processed[cs] = true;
        if (!(adj[ce].size() > 2)) processed[ce] = true;
        for (size_t k = 1; k + 1 < ids.size(); k++) processed[ids[k]] = true;
        std::vector<vec2> c;
        for (auto id : ids) c.push_back(nodes[id]);
        plg.add_polyline(c);
      }
      processed[i] = true;
    }
  delete g;
  plg.optimize();
}

}  // namespace n2
}  // namespace orc
