// Host replay of the PLGMatchesManager side effects of the path.
//
// The reference records every emitted chain while it runs: plgmm.add_matched_3dpolyline(chain)
// (src/edgegraph3d/matching/plg_matching/plg_matching_from_refpoints.cpp:74-77) turns each
// consecutive pair of chain points into a direct connection of the 3-D polyline graph
// (plg_matches_manager.cpp:110-116, polyline_graph_3d_hmap_impl.cpp:47-68,121-141) and marks the 2-D
// interval between their observations as matched on every view both points see
// (plg_matches_manager.cpp:99-108,118-173). On this path the manager is write-only, so the GPU
// never needs it; this file rebuilds it afterwards from the ordered edge-point cloud (the chain
// boundaries are in `key`), in one pass, with flat arrays instead of the reference's hash map of
// glm::vec3 / vector-of-vectors / std::set per polyline:
//   * node lookup: open-addressing table over the canonical bit pattern of (x,y,z) — the
//     reference's key equality is float ==, so -0 and +0 are one key and a NaN never matches;
//   * connections of a node: singly linked lists in two flat arrays (insertion order kept);
//   * matched intervals: the reference's std::set orders intervals by start.segment_index ONLY, so a
//     second interval starting on the same segment of the same polyline is silently dropped — kept
//     here as "first insertion wins" through a hash set of (polyline, start segment).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <unordered_set>
#include <vector>

#include "../../include/eg3d_host.h"

namespace {

const float kInvalid = -1.0f;  // INVALID_POINT_COORDS (polyline_graph_3d.hpp:57)

struct NodeTable {
  std::vector<uint32_t> slot;  // node id + 1, 0 = empty
  std::vector<float> key;      // [slots][3] the coordinates the slot was inserted with
  uint64_t mask = 0;
  explicit NodeTable(uint64_t n_points) {
    uint64_t cap = 64;
    while (cap < n_points * 2 + 16) cap <<= 1;
    slot.assign(cap, 0);
    key.assign(cap * 3, 0.f);
    mask = cap - 1;
  }
  static uint32_t canon(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u == 0x80000000u ? 0u : u;  // -0 == +0
  }
  static uint64_t hash(const float* X) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < 3; k++) {
      h ^= canon(X[k]);
      h *= 0xBF58476D1CE4E5B9ull;
      h ^= h >> 29;
    }
    return h;
  }
  // slot index holding X, or the empty slot where it would go (found = false)
  uint64_t find(const float* X, bool& found) const {
    found = false;
    if (X[0] != X[0] || X[1] != X[1] || X[2] != X[2]) {  // NaN: equal to nothing; park at any empty slot
      uint64_t i = hash(X) & mask;
      while (slot[i]) i = (i + 1) & mask;
      return i;
    }
    uint64_t i = hash(X) & mask;
    while (slot[i]) {
      const float* k = &key[i * 3];
      if (k[0] == X[0] && k[1] == X[1] && k[2] == X[2]) {
        found = true;
        return i;
      }
      i = (i + 1) & mask;
    }
    return i;
  }
};

struct Plp {
  uint32_t seg;
  float x, y;
};

// is_ordered_2dlinepoints (geometric_utilities.cpp:1375-1377)
bool ordered(float ax, float ay, float bx, float by, float cx, float cy) {
  return (bx - ax) * (cx - bx) > 0 || (by - ay) * (cy - by) > 0 || ((ax == bx && ay == by) || (bx == cx && by == cy));
}

}  // namespace

extern "C" int eg3d_host_replay_matches(const eg3d_scene* sc, const eg3d_edgepoints* pts, eg3d_graph3d* out) {
  if (!sc || !pts || !out) return -1;
  memset(out, 0, sizeof(*out));
  if (sc->n_views < 1 || !sc->view_pl_off || !sc->pl_vtx_off || !sc->vtx_xy) return -1;
  const int V = sc->n_views;
  const uint64_t N = pts->n_points;
  const uint32_t NP = sc->view_pl_off[V];
  if (N >= 0xfffffff0ull) return -3;  // node / polyline ids are 32-bit here (documented narrowing)
  // the cloud normally comes from eg3d_match_*; one built by the caller is checked before its ids index the scene
  if (N && (!pts->X || !pts->obs_off || !pts->key)) return -1;
  for (uint64_t p = 0; p < N; p++)
    if (pts->obs_off[p + 1] < pts->obs_off[p] || pts->obs_off[p + 1] > pts->n_obs) return -1;
  if (N && pts->obs_off[N] && (!pts->obs_view || !pts->obs_pl || !pts->obs_seg || !pts->obs_xy)) return -1;
  for (uint64_t o = 0, no = N ? pts->obs_off[N] : 0; o < no; o++) {
    const int32_t v = pts->obs_view[o];
    if (v < 0 || v >= V) return -1;
    const uint32_t npl = sc->view_pl_off[v + 1] - sc->view_pl_off[v];
    if (pts->obs_pl[o] >= npl) return -1;
    const uint32_t g = sc->view_pl_off[v] + pts->obs_pl[o];
    const uint32_t nv = sc->pl_vtx_off[g + 1] - sc->pl_vtx_off[g];
    if (nv < 2 || pts->obs_seg[o] >= nv - 1) return -1;  // a segment index of a polyline that has segments
  }
  NodeTable table(N);
  std::vector<float> node_X;
  std::vector<uint64_t> node_point;
  std::vector<uint32_t> head, tail, cnt;      // per node: connection list
  std::vector<uint32_t> link_pl, link_next;   // list cells
  std::vector<uint32_t> pl_s, pl_e;
  uint64_t real_nodes = 0;

  auto node_valid = [&](uint32_t id) { return node_X[3 * (size_t)id] != kInvalid && node_X[3 * (size_t)id + 1] != kInvalid; };
  auto new_node = [&](const float* X) {
    const uint32_t id = (uint32_t)node_point.size();
    node_X.insert(node_X.end(), X, X + 3);
    node_point.push_back(~0ull);
    head.push_back(~0u);
    tail.push_back(~0u);
    cnt.push_back(0);
    real_nodes++;
    return id;
  };
  // get_node_id (polyline_graph_3d_hmap_impl.cpp:47-68)
  auto node_of = [&](const float* X) {
    bool found;
    const uint64_t i = table.find(X, found);
    if (found) {
      const uint32_t id = table.slot[i] - 1;
      if (node_valid(id)) return id;
      // a node whose x or y equals INVALID_POINT_COORDS is "invalid": wiped, and a fresh node takes the key
      node_X[3 * (size_t)id] = node_X[3 * (size_t)id + 1] = node_X[3 * (size_t)id + 2] = kInvalid;
      head[id] = tail[id] = ~0u;
      cnt[id] = 0;
      const uint32_t nid = new_node(X);
      table.slot[i] = nid + 1;
      return nid;
    }
    const uint32_t id = new_node(X);
    table.slot[i] = id + 1;
    table.key[i * 3] = X[0];
    table.key[i * 3 + 1] = X[1];
    table.key[i * 3 + 2] = X[2];
    return id;
  };
  auto connect = [&](uint32_t node, uint32_t pl) {
    const uint32_t cell = (uint32_t)link_pl.size();
    link_pl.push_back(pl);
    link_next.push_back(~0u);
    if (tail[node] == ~0u)
      head[node] = cell;
    else
      link_next[tail[node]] = cell;
    tail[node] = cell;
    cnt[node]++;
  };
  // add_direct_connection(start, end) -> internal_add_polyline unless is_duplicate (:99-127)
  auto add_connection = [&](uint32_t a, uint32_t b) {
    const uint32_t scan = cnt[a] < cnt[b] ? a : b;  // "smallest_connections": s unless e is strictly smaller
    const uint32_t from = cnt[a] < cnt[b] ? a : b;
    (void)from;
    for (uint32_t c = head[scan]; c != ~0u; c = link_next[c]) {
      const uint32_t p = link_pl[c];
      // polyline equality: same extremes (either orientation) and the same two coordinates — which
      // are the extremes' node coordinates, compared with float == (a NaN coordinate never matches)
      const bool same = (pl_s[p] == a && pl_e[p] == b) || (pl_s[p] == b && pl_e[p] == a);
      if (!same) continue;
      const float* xa = &node_X[3 * (size_t)a];
      const float* xb = &node_X[3 * (size_t)b];
      if (xa[0] == xa[0] && xa[1] == xa[1] && xa[2] == xa[2] && xb[0] == xb[0] && xb[1] == xb[1] && xb[2] == xb[2]) return;
    }
    const uint32_t id = (uint32_t)pl_s.size();
    pl_s.push_back(a);
    pl_e.push_back(b);
    connect(a, id);
    if (a != b) connect(b, id);
  };

  // matched 2-D intervals
  struct Iv {
    uint32_t gpl, s_seg, e_seg;
    float sx, sy, ex, ey;
  };
  std::vector<Iv> ivs;
  std::unordered_set<uint64_t> iv_keys;
  auto insert_iv = [&](uint32_t gpl, const Plp& a, const Plp& b) {
    if (!iv_keys.insert(((uint64_t)gpl << 32) | a.seg).second) return;  // std::set keyed on start.segment_index
    ivs.push_back({gpl, a.seg, b.seg, a.x, a.y, b.x, b.y});
  };
  // add_matched_2dsegment (plg_matches_manager.cpp:99-108)
  auto add_2dsegment = [&](int view, uint32_t pl, const Plp& a, const Plp& b) {
    const uint32_t gpl = sc->view_pl_off[view] + pl;
    if (a.seg < b.seg)
      insert_iv(gpl, a, b);
    else if (a.seg > b.seg)
      insert_iv(gpl, b, a);
    else {
      const float* v = sc->vtx_xy + 2 * ((size_t)sc->pl_vtx_off[gpl] + a.seg);
      if (ordered(v[0], v[1], a.x, a.y, b.x, b.y))
        insert_iv(gpl, a, b);
      else
        insert_iv(gpl, b, a);
    }
  };

  std::vector<int64_t> slot1((size_t)V), slot2((size_t)V);  // observation index of each view in p1 / p2, -1 = unseen
  for (uint64_t i = 1; i < N; i++) {
    const uint32_t* k0 = pts->key + 4 * (i - 1);
    const uint32_t* k1 = pts->key + 4 * i;
    if (k0[0] != k1[0] || k0[1] != k1[1] || k0[2] != k1[2] || k1[3] != k0[3] + 1) continue;  // first point of a chain
    // ---- add_matched_3dsegment(p[i-1], p[i]) (plg_matches_manager.cpp:110-173)
    const uint32_t na = node_of(pts->X + 3 * (i - 1));
    const uint32_t nb = node_of(pts->X + 3 * i);
    add_connection(na, nb);
    node_point[na] = i - 1;  // set_observations: last writer wins (first, then second)
    node_point[nb] = i;
    std::fill(slot1.begin(), slot1.end(), -1);
    std::fill(slot2.begin(), slot2.end(), -1);
    for (uint64_t o = pts->obs_off[i - 1]; o < pts->obs_off[i]; o++) slot1[pts->obs_view[o]] = (int64_t)o;
    for (uint64_t o = pts->obs_off[i]; o < pts->obs_off[i + 1]; o++) slot2[pts->obs_view[o]] = (int64_t)o;
    for (int v = 0; v < V; v++) {
      if (slot1[v] < 0 || slot2[v] < 0) continue;
      const uint64_t o1 = (uint64_t)slot1[v], o2 = (uint64_t)slot2[v];
      const Plp a = {pts->obs_seg[o1], pts->obs_xy[2 * o1], pts->obs_xy[2 * o1 + 1]};
      const Plp b = {pts->obs_seg[o2], pts->obs_xy[2 * o2], pts->obs_xy[2 * o2 + 1]};
      const uint32_t pl1 = pts->obs_pl[o1], pl2 = pts->obs_pl[o2];
      if (pl1 == pl2) {
        add_2dsegment(v, pl1, a, b);
        continue;
      }
      // different polylines: only if the first point sits on an extreme of its polyline that the
      // second polyline shares (:132-164)
      const uint32_t g1 = sc->view_pl_off[v] + pl1, g2 = sc->view_pl_off[v] + pl2;
      const uint32_t n1 = sc->pl_vtx_off[g1 + 1] - sc->pl_vtx_off[g1];
      const float* v1 = sc->vtx_xy + 2 * (size_t)sc->pl_vtx_off[g1];
      if (n1 < 2) continue;
      uint32_t node_id = 0;
      bool extreme = false;
      if (a.seg == 0 && a.x == v1[0] && a.y == v1[1]) {  // is_start(plp)
        node_id = sc->pl_start[g1];
        extreme = true;
      }
      if (!extreme && a.seg == n1 - 2 && a.x == v1[2 * (n1 - 1)] && a.y == v1[2 * (n1 - 1) + 1]) {  // is_end(plp)
        node_id = sc->pl_end[g1];
        extreme = true;
      }
      if (!extreme) continue;
      const uint32_t n2 = sc->pl_vtx_off[g2 + 1] - sc->pl_vtx_off[g2];
      const float* v2 = sc->vtx_xy + 2 * (size_t)sc->pl_vtx_off[g2];
      if (n2 < 2) continue;
      Plp ext;
      if (node_id == sc->pl_start[g2])  // get_extreme_plp(node_id, valid) (polyline_graph_2d.cpp:151-160)
        ext = {0u, v2[0], v2[1]};
      else if (node_id == sc->pl_end[g2])
        ext = {n2 - 2, v2[2 * (n2 - 1)], v2[2 * (n2 - 1) + 1]};
      else
        continue;
      add_2dsegment(v, pl2, ext, b);
    }
  }

  // ---- flatten
  const uint64_t NN = node_point.size();
  out->n_nodes = NN;
  out->n_real_nodes = real_nodes;
  out->n_polylines = pl_s.size();
  out->n_scene_polylines = NP;
  auto dup = [](const void* src, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(p, src, bytes);
    return p;
  };
  out->node_X = (float*)dup(node_X.data(), sizeof(float) * node_X.size());
  out->node_point = (uint64_t*)dup(node_point.data(), sizeof(uint64_t) * NN);
  out->pl_start = (uint32_t*)dup(pl_s.data(), sizeof(uint32_t) * pl_s.size());
  out->pl_end = (uint32_t*)dup(pl_e.data(), sizeof(uint32_t) * pl_e.size());
  out->conn_off = (uint64_t*)malloc(sizeof(uint64_t) * (NN + 1));
  out->conn_pl = (uint32_t*)malloc(sizeof(uint32_t) * (link_pl.size() + 1));
  uint64_t w = 0;
  for (uint64_t n = 0; n < NN; n++) {
    out->conn_off[n] = w;
    for (uint32_t c = head[n]; c != ~0u; c = link_next[c]) out->conn_pl[w++] = link_pl[c];
  }
  out->conn_off[NN] = w;
  std::sort(ivs.begin(), ivs.end(), [](const Iv& a, const Iv& b) { return a.gpl != b.gpl ? a.gpl < b.gpl : a.s_seg < b.s_seg; });
  const uint64_t NI = ivs.size();
  out->iv_off = (uint64_t*)calloc((size_t)NP + 1, sizeof(uint64_t));
  out->iv_start_seg = (uint32_t*)malloc(sizeof(uint32_t) * (NI + 1));
  out->iv_end_seg = (uint32_t*)malloc(sizeof(uint32_t) * (NI + 1));
  out->iv_start_xy = (float*)malloc(sizeof(float) * 2 * (NI + 1));
  out->iv_end_xy = (float*)malloc(sizeof(float) * 2 * (NI + 1));
  for (uint64_t k = 0; k < NI; k++) {
    out->iv_off[ivs[k].gpl + 1]++;
    out->iv_start_seg[k] = ivs[k].s_seg;
    out->iv_end_seg[k] = ivs[k].e_seg;
    out->iv_start_xy[2 * k] = ivs[k].sx;
    out->iv_start_xy[2 * k + 1] = ivs[k].sy;
    out->iv_end_xy[2 * k] = ivs[k].ex;
    out->iv_end_xy[2 * k + 1] = ivs[k].ey;
  }
  for (uint32_t p = 0; p < NP; p++) out->iv_off[p + 1] += out->iv_off[p];
  return 0;
}

extern "C" void eg3d_host_free_graph3d(eg3d_graph3d* g) {
  if (!g) return;
  free(g->node_X);
  free(g->node_point);
  free(g->pl_start);
  free(g->pl_end);
  free(g->conn_off);
  free(g->conn_pl);
  free(g->iv_off);
  free(g->iv_start_seg);
  free(g->iv_end_seg);
  free(g->iv_start_xy);
  free(g->iv_end_xy);
  memset(g, 0, sizeof(*g));
}
