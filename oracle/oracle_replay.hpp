// ORACLE — TEST INFRASTRUCTURE ONLY. PARITY UNPINNED.
// CPU restatement of the PLGMatchesManager side effects of the path, kept in
// the reference's own shape (hash map of coordinates -> node id, vector-of-vectors connections,
// one std::set of intervals per polyline ordered by start.segment_index) so that it is an
// independent check of the product's flat-array replay (edgegraph3d_amd/host/replay.cpp).
// Follows, relative to /root/reference:
//   PLGMatchesManager::add_matched_3dpolyline / _3dsegment / _2dsegment
//       src/edgegraph3d/matching/plg_matching/plg_matches_manager.cpp:99-180
//   interval_compare                include/edgegraph3d/matching/plg_matching/plg_matches_manager.hpp:59-65
//   PolyLineGraph3DHMapImpl::get_node_id / internal_add_polyline / is_duplicate / add_direct_connection
//       src/edgegraph3d/plgs/polyline_graph_3d_hmap_impl.cpp:47-68,88-141
//   KeyFuncs3d (hash of the truncated coordinates, float == equality)
//       include/edgegraph3d/plgs/polyline_graph_3d_hmap_impl.hpp:64-75
//   PolyLineGraph3D::polyline::operator==, invalidate_node, is_valid_node, get_next_node_id, set_observations
//       src/edgegraph3d/plgs/polyline_graph_3d.cpp:264-267,367-375,461-467,477-480
//   polyline::is_start / is_end / get_extreme_plp(id, valid)   src/edgegraph3d/plgs/polyline_graph_2d.cpp:114-124,151-160
//   is_ordered_2dlinepoints         src/edgegraph3d/utils/geometry/geometric_utilities.cpp:1375-1377
#pragma once
#include <cstdint>
#include <set>
#include <unordered_map>
#include <vector>

#include "oracle_plg.hpp"

namespace orc {

struct vec3r {
  float x, y, z;
};
struct KeyFuncs3d {
  size_t operator()(const vec3r& k) const {
    // std::hash<int>()(float): the float is converted to int first (NaN / huge values are UB in the
    // reference; any value serves here because equality decides)
    auto h = [](float f) -> size_t {
      if (!(f > -2.0e9f && f < 2.0e9f)) return 0;
      return std::hash<int>()((int)f);
    };
    return h(k.x) ^ h(k.y) ^ h(k.z);
  }
  bool operator()(const vec3r& a, const vec3r& b) const { return a.x == b.x && a.y == b.y && a.z == b.z; }
};

struct pl_point_r {
  unsigned long segment_index;
  vec2 coords;
};
struct pl_interval_r {
  pl_point_r start, end;
};
struct interval_compare_r {
  bool operator()(const pl_interval_r& a, const pl_interval_r& b) const {
    return a.start.segment_index < b.start.segment_index;
  }
};
typedef std::set<pl_interval_r, interval_compare_r> sorted_pl_intervals_set;

struct PLG3D {
  struct polyline3 {
    unsigned long start, end;
    std::vector<vec3r> coords;
  };
  std::vector<polyline3> polylines;
  std::vector<std::vector<unsigned long>> connections;
  std::vector<vec3r> nodes_coords;
  std::vector<uint64_t> node_point;  // which edge-point's observations the node carries (set_observations)
  std::unordered_map<vec3r, unsigned long, KeyFuncs3d, KeyFuncs3d> point_map;
  unsigned long next_node_id = 0, real_nodes_amount = 0;

  bool is_valid_node(unsigned long id) const { return nodes_coords[id].x != -1.0f && nodes_coords[id].y != -1.0f; }
  void invalidate_node(unsigned long id) {
    nodes_coords[id] = vec3r{-1.0f, -1.0f, -1.0f};
    connections[id].clear();
  }
  unsigned long get_node_id(const vec3r& p) {
    auto it = point_map.find(p);
    if (it != point_map.end() && !is_valid_node(it->second)) {
      invalidate_node(it->second);
      it = point_map.end();
    }
    if (it == point_map.end()) {
      unsigned long id = next_node_id++;
      point_map[p] = id;
      connections.push_back(std::vector<unsigned long>());
      node_point.push_back(~0ull);
      nodes_coords.push_back(p);
      real_nodes_amount++;
      return id;
    }
    return it->second;
  }
  static bool vec_eq(const std::vector<vec3r>& a, const std::vector<vec3r>& b, bool inv) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++) {
      const vec3r& u = a[i];
      const vec3r& v = inv ? b[b.size() - 1 - i] : b[i];
      if (!(u.x == v.x && u.y == v.y && u.z == v.z)) return false;
    }
    return true;
  }
  bool pl_equal(const polyline3& a, const polyline3& b) const {
    return (a.start == b.start && a.end == b.end && vec_eq(a.coords, b.coords, false)) ||
           (a.start == b.end && a.end == b.start && vec_eq(a.coords, b.coords, true));
  }
  bool is_duplicate(const polyline3& pl) const {
    const auto& s = connections[pl.start];
    const auto& e = connections[pl.end];
    const auto& smallest = s.size() < e.size() ? s : e;
    for (auto id : smallest)
      if (pl_equal(polylines[id], pl)) return true;
    return false;
  }
  void add_direct_connection(const vec3r& a, const vec3r& b, unsigned long& ia, unsigned long& ib) {
    ia = get_node_id(a);
    ib = get_node_id(b);
    polyline3 pl{ia, ib, {nodes_coords[ia], nodes_coords[ib]}};
    if (!is_duplicate(pl)) {
      unsigned long id = polylines.size();
      polylines.push_back(pl);
      connections[pl.start].push_back(id);
      if (pl.start != pl.end) connections[pl.end].push_back(id);
    }
  }
};

inline bool is_ordered_2dlinepoints_r(const vec2& a, const vec2& b, const vec2& c) {
  return (b.x - a.x) * (c.x - b.x) > 0 || (b.y - a.y) * (c.y - b.y) > 0 ||
         ((a.x == b.x && a.y == b.y) || (b.x == c.x && b.y == c.y));
}

struct MatchesManager {
  Scene& sc;
  PLG3D plg3d;
  std::vector<std::vector<sorted_pl_intervals_set>> matched;  // [view][polyline]
  explicit MatchesManager(Scene& s) : sc(s) {
    for (auto& g : sc.plgs) matched.push_back(std::vector<sorted_pl_intervals_set>(g.polylines.size()));
  }
  void add_matched_2dsegment(int plg_id, unsigned long pl_id, const pl_point_r& a, const pl_point_r& b) {
    auto& S = matched[plg_id][pl_id];
    if (a.segment_index < b.segment_index)
      S.insert(pl_interval_r{a, b});
    else if (a.segment_index > b.segment_index)
      S.insert(pl_interval_r{b, a});
    else if (is_ordered_2dlinepoints_r(sc.plgs[plg_id].polylines[pl_id].polyline_coords[a.segment_index], a.coords, b.coords))
      S.insert(pl_interval_r{a, b});
    else
      S.insert(pl_interval_r{b, a});
  }
  // one chain point = (X, observations, view ids)
  struct P3 {
    vec3r X;
    std::vector<plg_point> obs;
    std::vector<int> views;
    uint64_t index;
  };
  void add_matched_3dsegment(const P3& p1, const P3& p2) {
    unsigned long n1, n2;
    plg3d.add_direct_connection(p1.X, p2.X, n1, n2);
    plg3d.node_point[n1] = p1.index;
    plg3d.node_point[n2] = p2.index;
    const size_t V = sc.plgs.size();
    std::vector<plg_point> plgps1(V), plgps2(V);
    std::vector<bool> seen1(V, false), seen2(V, false);
    for (size_t i = 0; i < p1.obs.size(); i++) {
      seen1[p1.views[i]] = true;
      plgps1[p1.views[i]] = p1.obs[i];
    }
    for (size_t i = 0; i < p2.obs.size(); i++) {
      seen2[p2.views[i]] = true;
      plgps2[p2.views[i]] = p2.obs[i];
    }
    for (size_t v = 0; v < V; v++)
      if (seen1[v] && seen2[v]) {
        const pl_point_r a{plgps1[v].plp.segment_index, plgps1[v].plp.coords};
        const pl_point_r b{plgps2[v].plp.segment_index, plgps2[v].plp.coords};
        if (plgps1[v].polyline_id != plgps2[v].polyline_id) {
          const polyline& pl = sc.plgs[v].polylines[plgps1[v].polyline_id];
          const auto& pc = pl.polyline_coords;
          if (pc.size() < 2) continue;
          unsigned long node_id = 0;
          bool is_extreme = false;
          if (a.segment_index == 0 && a.coords.x == pc[0].x && a.coords.y == pc[0].y) {
            node_id = pl.start;
            is_extreme = true;
          }
          if (!is_extreme && a.segment_index == pc.size() - 2 && a.coords.x == pc[pc.size() - 1].x &&
              a.coords.y == pc[pc.size() - 1].y) {
            node_id = pl.end;
            is_extreme = true;
          }
          if (is_extreme) {
            const polyline& next_pl = sc.plgs[v].polylines[plgps2[v].polyline_id];
            const auto& nc = next_pl.polyline_coords;
            if (nc.size() < 2) continue;
            bool valid = false;
            pl_point_r ext{0, vec2()};
            if (node_id == next_pl.start) {
              valid = true;
              ext = pl_point_r{0, nc[0]};
            } else if (node_id == next_pl.end) {
              valid = true;
              ext = pl_point_r{nc.size() - 2, nc[nc.size() - 1]};
            }
            if (valid) add_matched_2dsegment((int)v, plgps2[v].polyline_id, ext, b);
          }
        } else
          add_matched_2dsegment((int)v, plgps1[v].polyline_id, a, b);
      }
  }
  void add_matched_3dpolyline(const std::vector<P3>& pl) {
    for (size_t i = 1; i < pl.size(); i++) add_matched_3dsegment(pl[i - 1], pl[i]);
  }
};

}  // namespace orc
