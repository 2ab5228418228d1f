// Edge image -> polyline graph: the one-off host step that produces the per-view
// PolyLineGraph2D the hot path consumes.
//
// Behaviour reproduced (reference): convertEdgeImagePolyLineGraph_optimized =
//   convertEdgeImagePixelToGraph_NoCycles + convert_EdgeGraph_to_PolyLineGraph + plg.optimize()
//   (src/edgegraph3d/io/input/convert_edge_images_pixel_to_segment.cpp:294-426, 428-626, 868-883;
//    src/edgegraph3d/plgs/polyline_graph_2d_hmap_impl.cpp:47-266;
//    src/edgegraph3d/plgs/polyline_graph_2d.cpp:76-98, 905-1160, 1315-1355, 1926-2066;
//    src/edgegraph3d/plgs/graph_adjacency_set_no_type.cpp:73-150).
// Polyline ids and node ids are the reference's vector positions; invalidated polylines keep their
// id and lose their vertices.
//
// Own design (the oracle keeps the reference's containers, this file does not):
//   * pixel graph: a pixel has at most 8 neighbours, so adjacency is a fixed [n][8] table kept sorted
//     ascending (the reference iterates std::set<ulong>) with a count byte; the bounded reachability
//     test walks two explicit LIFO frontiers over a byte array of marks;
//   * polyline vertices live in ONE append-only pool (a polyline = offset + count; simplifying,
//     merging or splitting appends the new run), node coordinates are SoA, the coordinate -> node map
//     is an open-addressing table over the coordinate bits;
//   * components are labelled with an explicit stack into flat arrays; the per-component polyline
//     sets of the smooth-length filter are sorted id runs of one array.
// Two behaviours of the reference are kept on purpose because they decide the output (see
// oracle/oracle_n2.hpp U1, U2): out-of-image reads of the "useless hub" test address the image as
// one row-major buffer, and the reachability test's marks are never reset (its reset loop runs
// over a vector<bool>), so marks accumulate over the whole image.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/eg3d_host.h"

namespace {

const float kInvalid = -1.0f;

// ---- float helpers: same operations, same order as the reference's geometric utilities ----------
inline float sqdist(float ax, float ay, float bx, float by) {
  // squared_2d_distance goes through pow(float, 2): evaluated in double, rounded once (Q5)
  const double dx = (double)(ax - bx), dy = (double)(ay - by);
  return (float)(dx * dx + dy * dy);
}
inline float dist(float ax, float ay, float bx, float by) { return std::sqrt(sqdist(ax, ay, bx, by)); }

struct Line {
  float a, b, c;
};
inline Line line_through(float ax, float ay, float bx, float by) {
  if (ax == bx) return Line{1.0f, 0.0f, -ax};
  const float m = (by - ay) / (bx - ax);
  const float q = ay - m * ax;
  return Line{m, -1.0f, q};
}
inline float dist_point_line_sq(float px, float py, const Line& l) {
  float den = l.a * px + l.b * py + l.c;
  den *= den;
  return den / (l.a * l.a + l.b * l.b);
}

// ---- pixel graph ------------------------------------------------------------------------------------
struct PixelGraph {
  std::vector<uint32_t> adj;  // [n][8] ascending
  std::vector<uint8_t> deg, mark;
  std::vector<uint32_t> cur, next;
  explicit PixelGraph(size_t n) : adj(n * 8), deg(n, 0), mark(n, 0) {}
  void insert(uint32_t a, uint32_t b) {
    uint32_t* row = &adj[(size_t)a * 8];
    int k = deg[a];
    for (int i = 0; i < k; i++)
      if (row[i] == b) return;
    while (k > 0 && row[k - 1] > b) {
      row[k] = row[k - 1];
      k--;
    }
    row[k] = b;
    deg[a]++;
  }
  void add_edge(uint32_t a, uint32_t b) {
    insert(a, b);
    insert(b, a);
  }
  // is `to` within max_dist+1 steps of `from`, not walking through marked nodes? Marks persist (U2):
  // the only entries the reference ever clears are 0 and 1, after every level.
  bool reachable(uint32_t from, uint32_t to, unsigned max_dist) {
    bool clear0 = false, clear1 = false;
    (from != 0 ? clear1 : clear0) = true;
    mark[from] = 1;
    cur.clear();
    cur.push_back(from);
    bool found = false;
    for (unsigned level = 0; level <= max_dist && !cur.empty(); level++) {
      next.clear();
      while (!found && !cur.empty()) {
        const uint32_t n = cur.back();
        cur.pop_back();
        const uint32_t* row = &adj[(size_t)n * 8];
        for (int i = 0; i < deg[n]; i++) {
          const uint32_t c = row[i];
          if (c == to) {
            found = true;
            break;
          }
          if (!mark[c]) {
            next.push_back(c);
            mark[c] = 1;
            (c != 0 ? clear1 : clear0) = true;
          }
        }
      }
      if (clear0 && !mark.empty()) mark[0] = 0;
      if (clear1 && mark.size() > 1) mark[1] = 0;
      cur.swap(next);  // the reference copies the next stack: same LIFO order
    }
    return found;
  }
};

// ---- polyline graph -----------------------------------------------------------------------------------
struct Poly {
  uint32_t start, end;
  uint32_t off, n;  // vertex run in the pool
  float length;
};

struct Graph {
  std::vector<Poly> pls;
  std::vector<float> pool;  // x,y pairs
  std::vector<std::vector<uint32_t>> conn;
  std::vector<float> nx, ny;
  // coordinate -> node, open addressing
  std::vector<uint32_t> slot;
  std::vector<float> kx, ky;
  uint64_t mask = 0;

  void init_table(size_t expected) {
    uint64_t cap = 64;
    while (cap < expected * 2 + 16) cap <<= 1;
    slot.assign(cap, 0);
    kx.assign(cap, 0.f);
    ky.assign(cap, 0.f);
    mask = cap - 1;
  }
  static uint64_t hash(float x, float y) {
    uint32_t a, b;
    memcpy(&a, &x, 4);
    memcpy(&b, &y, 4);
    if (a == 0x80000000u) a = 0;
    if (b == 0x80000000u) b = 0;
    uint64_t h = ((uint64_t)a << 32 | b) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 31);
  }
  void grow() {
    std::vector<uint32_t> os;
    std::vector<float> ox, oy;
    os.swap(slot);
    ox.swap(kx);
    oy.swap(ky);
    init_table(os.size());
    for (size_t i = 0; i < os.size(); i++)
      if (os[i] && os[i] != ~0u) {
        uint64_t j = hash(ox[i], oy[i]) & mask;
        while (slot[j]) j = (j + 1) & mask;
        slot[j] = os[i];
        kx[j] = ox[i];
        ky[j] = oy[i];
      }
    used = 0;
    for (auto s : slot) used += s != 0;
  }
  size_t used = 0;
  // slot of (x,y) or of the free place for it; tombstones (~0) are skipped when searching
  uint64_t find(float x, float y, bool& found) const {
    found = false;
    uint64_t i = hash(x, y) & mask, first_free = ~0ull;
    while (slot[i]) {
      if (slot[i] == ~0u) {
        if (first_free == ~0ull) first_free = i;
      } else if (kx[i] == x && ky[i] == y) {
        found = true;
        return i;
      }
      i = (i + 1) & mask;
    }
    return first_free != ~0ull ? first_free : i;
  }
  void erase_key(float x, float y) {
    bool f;
    const uint64_t i = find(x, y, f);
    if (f) slot[i] = ~0u;
  }

  const float* vtx(const Poly& p) const { return &pool[2 * (size_t)p.off]; }
  bool node_valid(uint32_t n) const { return nx[n] != kInvalid && ny[n] != kInvalid; }
  bool poly_valid(uint32_t id) const {
    const Poly& p = pls[id];
    if (!node_valid(p.start) || !node_valid(p.end) || p.n <= 1) return false;
    const float* v = vtx(p);
    return nx[p.start] == v[0] && ny[p.start] == v[1] && nx[p.end] == v[2 * (p.n - 1)] && ny[p.end] == v[2 * (p.n - 1) + 1];
  }
  static float run_length(const float* v, uint32_t n) {
    float len = 0.0f;
    for (uint32_t i = 1; i < n; i++) len += dist(v[2 * i], v[2 * i + 1], v[2 * i - 2], v[2 * i - 1]);
    return len;
  }
  uint32_t other_end(const Poly& p, uint32_t e) const { return e == p.start ? p.end : p.start; }

  void wipe_node(uint32_t n) {  // PolyLineGraph2D::invalidate_node
    nx[n] = ny[n] = kInvalid;
    const std::vector<uint32_t> snapshot = conn[n];  // empty whenever this is reached (see oracle_n2.hpp)
    for (uint32_t pid : snapshot) remove_poly(pid);
    conn[n].clear();
  }
  void wipe_node_and_key(uint32_t n) {  // PolyLineGraph2DHMapImpl::invalidate_node
    erase_key(nx[n], ny[n]);
    wipe_node(n);
  }
  void drop_connection(uint32_t n, uint32_t pid) {
    std::vector<uint32_t>& c = conn[n];
    c.erase(std::remove(c.begin(), c.end(), pid), c.end());
    if (c.empty()) wipe_node(n);
  }
  void remove_poly(uint32_t pid) {
    const uint32_t s = pls[pid].start, e = pls[pid].end;
    drop_connection(s, pid);
    drop_connection(e, pid);
    pls[pid].n = 0;
    pls[pid].length = -1.0f;  // INVALID_POLYLINE_LENGTH
  }
  uint32_t node_of(float x, float y) {
    if (used * 2 + 16 > slot.size()) grow();
    bool found;
    uint64_t i = find(x, y, found);
    if (found) {
      const uint32_t id = slot[i] - 1;
      if (node_valid(id)) return id;
      wipe_node_and_key(id);  // erases the key of its (already wiped) coordinates, not this one
      // fall through: a fresh node takes this key
      const uint32_t nid = (uint32_t)nx.size();
      slot[i] = nid + 1;
      conn.emplace_back();
      nx.push_back(x);
      ny.push_back(y);
      return nid;
    }
    const uint32_t id = (uint32_t)nx.size();
    if (slot[i] == 0) used++;
    slot[i] = id + 1;
    kx[i] = x;
    ky[i] = y;
    conn.emplace_back();
    nx.push_back(x);
    ny.push_back(y);
    return id;
  }
  static bool runs_equal(const float* a, uint32_t na, const float* b, uint32_t nb, bool inv) {
    if (na != nb) return false;
    for (uint32_t i = 0; i < na; i++) {
      const uint32_t j = inv ? na - 1 - i : i;
      if (!(a[2 * i] == b[2 * j] && a[2 * i + 1] == b[2 * j + 1])) return false;
    }
    return true;
  }
  bool same_poly(const Poly& a, const Poly& b) const {
    return (a.start == b.start && a.end == b.end && runs_equal(vtx(a), a.n, vtx(b), b.n, false)) ||
           (a.start == b.end && a.end == b.start && runs_equal(vtx(a), a.n, vtx(b), b.n, true));
  }
  void add_poly(const Poly& p) {  // internal_add_polyline
    const std::vector<uint32_t>& s = conn[p.start];
    const std::vector<uint32_t>& e = conn[p.end];
    const std::vector<uint32_t>& smallest = s.size() < e.size() ? s : e;
    for (uint32_t id : smallest)
      if (same_poly(pls[id], p)) return;
    const uint32_t id = (uint32_t)pls.size();
    pls.push_back(p);
    conn[p.start].push_back(id);
    if (p.start != p.end) conn[p.end].push_back(id);
  }
  // add_polyline(coords) incl. filter_polyline; v = n (x,y) pairs (may point into the pool: copied first)
  void add_run(const float* v_in, uint32_t n) {
    std::vector<float> v(v_in, v_in + 2 * (size_t)n);
    if (n == 4 && v[0] == v[6] && v[1] == v[7] && sqdist(v[2], v[3], v[4], v[5]) <= 4) {
      const float mx = (v[2] + v[4]) / 2, my = (v[3] + v[5]) / 2;
      v.resize(4);
      v[2] = mx;
      v[3] = my;
      n = 2;
    }
    Poly p;
    p.start = node_of(v[0], v[1]);
    p.end = node_of(v[2 * (n - 1)], v[2 * (n - 1) + 1]);
    p.off = (uint32_t)(pool.size() / 2);
    p.n = n;
    pool.insert(pool.end(), v.begin(), v.end());
    p.length = run_length(vtx(p), n);
    add_poly(p);
  }
  void connect_nodes(uint32_t a, uint32_t b) {  // add_direct_connection
    Poly p;
    p.start = a;
    p.end = b;
    p.off = (uint32_t)(pool.size() / 2);
    p.n = 2;
    const float v[4] = {nx[a], ny[a], nx[b], ny[b]};
    pool.insert(pool.end(), v, v + 4);
    p.length = run_length(vtx(p), 2);
    add_poly(p);
  }
  bool is_extreme(uint32_t n) const { return conn[n].size() == 1 && pls[conn[n][0]].start != pls[conn[n][0]].end; }
};

// simplify_polyline (polyline_graph_2d.cpp:905-1013) on a vertex run; returns the kept vertex indices
bool linearizable(const float* v, size_t s, size_t e, float maxsq) {
  const Line l = line_through(v[2 * s], v[2 * s + 1], v[2 * e], v[2 * e + 1]);
  for (size_t i = s + 1; i < e; i++)
    if (dist_point_line_sq(v[2 * i], v[2 * i + 1], l) > maxsq) return false;
  return true;
}
void simplify_run(const float* v, uint32_t n, float max_dist, std::vector<uint32_t>& keep) {
  const float maxsq = max_dist * max_dist;
  size_t start = 0, end = n - 1;
  std::vector<uint32_t> tail;
  keep.clear();
  keep.push_back((uint32_t)start);
  tail.push_back((uint32_t)end);
  while (end > start + 1) {
    size_t se, eb, max_se = end, min_eb = start;
    for (;;) {
      // find_max_se
      if (max_se <= start)
        se = start;
      else {
        se = start + 1;
        for (size_t c = max_se; c > start + 1; c--)
          if (linearizable(v, start, c, maxsq)) {
            se = c;
            break;
          }
      }
      if (se == end) {
        eb = 0;
        break;
      }
      // find_min_eb
      if (min_eb >= end)
        eb = end;
      else {
        eb = end - 1;
        for (size_t c = min_eb; c < end - 1; c++)
          if (linearizable(v, c, end, maxsq)) {
            eb = c;
            break;
          }
      }
      max_se--;
      min_eb++;
      if (!(eb < se)) break;
    }
    if (se == end) break;
    keep.push_back((uint32_t)se);
    if (se != eb) tail.push_back((uint32_t)eb);
    start = se;
    end = eb;
  }
  for (size_t k = tail.size(); k-- > 0;) keep.push_back(tail[k]);
}

struct Builder {
  Graph g;

  void simplify_all() {
    const size_t n = g.pls.size();
    std::vector<uint32_t> keep;
    for (size_t i = 0; i < n; i++)
      if (g.poly_valid((uint32_t)i)) {
        Poly& p = g.pls[i];
        simplify_run(g.vtx(p), p.n, 1.0f, keep);  // MAXIMUM_LINEARIZABILITY_DISTANCE
        const uint32_t off = (uint32_t)(g.pool.size() / 2);
        const size_t src = 2 * (size_t)p.off;
        for (uint32_t k : keep) {
          g.pool.push_back(g.pool[src + 2 * k]);
          g.pool.push_back(g.pool[src + 2 * k + 1]);
        }
        p.off = off;
        p.n = (uint32_t)keep.size();
        p.length = Graph::run_length(g.vtx(p), p.n);
      }
  }
  void remove_invalid() {
    const size_t n = g.pls.size();
    for (size_t i = 0; i < n; i++)
      if (!g.poly_valid((uint32_t)i)) g.remove_poly((uint32_t)i);
  }
  void remove_degenerate_loops() {
    const size_t n = g.pls.size();
    for (size_t i = 0; i < n; i++)
      if (g.poly_valid((uint32_t)i)) {
        const Poly& p = g.pls[i];
        const float* v = g.vtx(p);
        if (p.start == p.end || (v[0] == v[2 * (p.n - 1)] && v[1] == v[2 * (p.n - 1) + 1]))
          if (p.n < 5) g.remove_poly((uint32_t)i);
      }
  }
  void merge_two_connection_nodes() {
    for (uint32_t node = 0; node < g.conn.size(); node++)
      if (g.conn[node].size() == 2) {
        const uint32_t id1 = g.conn[node][0], id2 = g.conn[node][1];
        const Poly p1 = g.pls[id1], p2 = g.pls[id2];
        if (Graph::runs_equal(g.vtx(p1), p1.n, g.vtx(p2), p2.n, false) || Graph::runs_equal(g.vtx(p1), p1.n, g.vtx(p2), p2.n, true)) {
          g.remove_poly(id2);
          continue;
        }
        if (g.other_end(p1, node) != node && g.other_end(p2, node) != node) {
          // merge_polylines: the run of p3 in the four orientations
          std::vector<float> v;
          auto app = [&](const Poly& p, bool rev, bool skip_first) {
            const float* s = &g.pool[2 * (size_t)p.off];
            for (uint32_t k = skip_first ? 1 : 0; k < p.n; k++) {
              const uint32_t i = rev ? p.n - 1 - k : k;
              v.push_back(s[2 * i]);
              v.push_back(s[2 * i + 1]);
            }
          };
          Poly p3;
          if (p1.start == p2.start) {
            p3.start = p1.end;
            p3.end = p2.end;
            app(p1, true, false);
            app(p2, false, true);
          } else if (p1.start == p2.end) {
            p3.start = p2.start;
            p3.end = p1.end;
            app(p2, false, false);
            app(p1, false, true);
          } else if (p1.end == p2.start) {
            p3.start = p1.start;
            p3.end = p2.end;
            app(p1, false, false);
            app(p2, false, true);
          } else {
            p3.start = p1.start;
            p3.end = p2.start;
            app(p1, false, false);
            app(p2, true, true);
          }
          p3.off = (uint32_t)(g.pool.size() / 2);
          p3.n = (uint32_t)(v.size() / 2);
          g.pool.insert(g.pool.end(), v.begin(), v.end());
          p3.length = Graph::run_length(g.vtx(p3), p3.n);
          g.add_poly(p3);
          g.remove_poly(id1);
          g.remove_poly(id2);
          g.wipe_node_and_key(node);
        }
      }
  }
  // component label of every node + size of every component (compute_components' DFS order is irrelevant
  // to the labels: a component is numbered by its lowest node id)
  void components(std::vector<uint32_t>& label, std::vector<uint32_t>& size) const {
    const size_t N = g.nx.size();
    label.assign(N, ~0u);
    size.clear();
    std::vector<uint32_t> stack;
    for (uint32_t s = 0; s < N; s++)
      if (label[s] == ~0u) {
        const uint32_t id = (uint32_t)size.size();
        uint32_t cnt = 0;
        stack.push_back(s);
        label[s] = id;
        while (!stack.empty()) {
          const uint32_t n = stack.back();
          stack.pop_back();
          cnt++;
          for (uint32_t pid : g.conn[n]) {
            const uint32_t o = g.other_end(g.pls[pid], n);
            if (label[o] == ~0u) {
              label[o] = id;
              stack.push_back(o);
            }
          }
        }
        size.push_back(cnt);
      }
  }
  bool crosses_any_polyline(float ax, float ay, float bx, float by) const {
    for (size_t i = 0; i < g.pls.size(); i++)
      if (g.poly_valid((uint32_t)i)) {
        const Poly& p = g.pls[i];
        const float* v = g.vtx(p);
        for (uint32_t k = 1; k < p.n; k++) {
          // intersect_segment_segment(polyline segment (v[k], v[k-1]), query): the query against the LINE of
          // the polyline's segment, then inside that segment's bounding box
          const float sx0 = v[2 * k], sy0 = v[2 * k + 1], sx1 = v[2 * k - 2], sy1 = v[2 * k - 1];
          const Line l = line_through(sx0, sy0, sx1, sy1);
          const float dx = bx - ax, dy = by - ay;
          const float num = l.a * ax + l.b * ay + l.c;
          const float den = l.a * dx + l.b * dy;
          if (den != 0) {
            const float t = -num / den;
            if (t >= 0 && t <= 1) {
              const float ix = ax + t * dx, iy = ay + t * dy;
              if (((sx0 <= ix && ix <= sx1) || (sx1 <= ix && ix <= sx0)) && ((sy0 <= iy && iy <= sy1) || (sy1 <= iy && iy <= sy0)))
                return true;
            }
          }
        }
      }
    return false;
  }
  void connect_close_extremes() {
    std::vector<uint32_t> ids;
    std::vector<float> px, py;
    for (uint32_t n = 0; n < g.nx.size(); n++)
      if (g.node_valid(n) && g.is_extreme(n)) {
        ids.push_back(n);
        px.push_back(g.nx[n]);
        py.push_back(g.ny[n]);
      }
    // reciprocal nearest neighbours no further than DIRECT_CONNECTION_EXTREMES_MAXDIST = 6 px; the first
    // minimal j wins ties (find_closest_pairs_with_max_dist, polyline_graph_2d.cpp:1315-1355)
    const size_t m = ids.size();
    std::vector<uint32_t> closest(m, ~0u);
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    for (size_t i = 0; i < m; i++) {
      float best = std::numeric_limits<float>::max();
      uint32_t bj = ~0u;
      for (size_t j = 0; j < m; j++)
        if (j != i) {
          const float d = sqdist(px[i], py[i], px[j], py[j]);
          if (d < best) {
            best = d;
            bj = (uint32_t)j;
          }
        }
      closest[i] = bj;
      if (bj != ~0u && bj < i && closest[bj] == i && sqdist(px[i], py[i], px[bj], py[bj]) <= 36.0f)
        pairs.emplace_back(ids[i], ids[bj]);
    }
    std::vector<uint32_t> label, size;
    components(label, size);
    // merging relabels the SMALLER component (by its node count at labelling time, as the reference's
    // std::set sizes, which it never updates either)
    std::vector<std::vector<uint32_t>> members(size.size());
    for (uint32_t n = 0; n < label.size(); n++) members[label[n]].push_back(n);
    for (const auto& pp : pairs)
      if (label[pp.first] != label[pp.second])
        if (!crosses_any_polyline(g.nx[pp.first], g.ny[pp.first], g.nx[pp.second], g.ny[pp.second])) {
          g.connect_nodes(pp.first, pp.second);
          const uint32_t la = label[pp.first], lb = label[pp.second];
          const uint32_t keep = members[la].size() < members[lb].size() ? lb : la;
          const uint32_t change = keep == la ? lb : la;
          for (uint32_t n : members[change]) label[n] = keep;
        }
  }
  void split_loops() {
    // split_loop walks from the loop's start plp "towards p.end", which for a loop IS p.start: the walk
    // (next_pl_point_by_length, polyline_graph_2d.cpp:455-470) covers no segment and reports the extreme
    // reached unless half the length is 0 — so no loop of length >= MINSPLITLOOP_LENGTH is ever split.
    // Kept as an explicit no-op with the reference's guard so that the stage order reads the same.
    const size_t n = g.pls.size();
    for (size_t i = 0; i < n; i++)
      if (g.poly_valid((uint32_t)i)) {
        const Poly& p = g.pls[i];
        if (p.length >= 10 && p.start == p.end) {
          const float half = p.length / 2;
          const bool walk_moves = 0.0f >= half;  // curlen (= 0) >= length
          (void)walk_moves;                     // never true here: half >= 5
        }
      }
  }
  static float max_smooth_length(const float* v, uint32_t n) {
    float maxl = 0.0f;
    uint32_t i = 1;
    while (i < n) {
      float cur = dist(v[2 * i], v[2 * i + 1], v[2 * i - 2], v[2 * i - 1]);
      for (i++; i < n; i++) {
        const float ax = v[2 * i] - v[2 * i - 2], ay = v[2 * i + 1] - v[2 * i - 1];
        const float bx = v[2 * i - 2] - v[2 * i - 4], by = v[2 * i - 1] - v[2 * i - 3];
        const float c = (ax * bx + ay * by) / std::sqrt((ax * ax + ay * ay) * (bx * bx + by * by));
        if (c)  // (sic) the cosine is used as a truth value: only an exact right angle ends a smooth section
          cur += dist(v[2 * i], v[2 * i + 1], v[2 * i - 2], v[2 * i - 1]);
        else
          break;
      }
      maxl = maxl < cur ? cur : maxl;
    }
    return maxl;
  }
  void filter_components_by_smooth_length() {
    std::vector<uint32_t> label, size;
    components(label, size);
    const size_t NP = g.pls.size();
    if (!NP) return;
    std::vector<float> smooth(NP, 0.0f);
    for (size_t i = 0; i < NP; i++)
      if (g.poly_valid((uint32_t)i)) smooth[i] = max_smooth_length(g.vtx(g.pls[i]), g.pls[i].n);
    std::vector<float> sorted = smooth;
    const size_t idx = (size_t)(NP * 0.82);  // TOP_FILTER_BY_POLYLINESMOOTHLENGTH
    std::nth_element(sorted.begin(), sorted.begin() + idx, sorted.end());
    const float threshold = sorted[idx];
    // a component survives if one of the polylines connected to its valid nodes reaches the threshold
    std::vector<uint8_t> keep(size.size(), 0);
    for (uint32_t n = 0; n < g.nx.size(); n++)
      if (g.node_valid(n))
        for (uint32_t pid : g.conn[n])
          if (smooth[pid] >= threshold) keep[label[n]] = 1;
    // removal in ascending polyline id per component, components ascending
    std::vector<std::pair<uint32_t, uint32_t>> doomed;  // (component, polyline)
    for (uint32_t n = 0; n < g.nx.size(); n++)
      if (g.node_valid(n) && !keep[label[n]])
        for (uint32_t pid : g.conn[n]) doomed.emplace_back(label[n], pid);
    std::sort(doomed.begin(), doomed.end());
    doomed.erase(std::unique(doomed.begin(), doomed.end()), doomed.end());
    for (const auto& d : doomed) g.remove_poly(d.second);
  }
  void optimize() {
    remove_invalid();
    remove_degenerate_loops();
    merge_two_connection_nodes();
    simplify_all();
    connect_close_extremes();
    simplify_all();
    split_loops();
    filter_components_by_smooth_length();
  }
};

}  // namespace

extern "C" int eg3d_plg_build_from_mask(const uint8_t* mask_in, int width, int height, eg3d_plg_view* out) {
  if (!mask_in || !out || width <= 0 || height <= 0) return -1;
  memset(out, 0, sizeof(*out));
  const int rows = height, cols = width;
  const long total = (long)rows * cols;
  std::vector<uint8_t> mask((size_t)total);
  for (long i = 0; i < total; i++) mask[i] = mask_in[i] ? 1 : 0;
  auto edge = [&](int i, int j) -> bool {  // U1: one row-major buffer, nothing outside it
    const long idx = (long)i * cols + j;
    return idx >= 0 && idx < total && mask[idx] != 0;
  };
  // ---- nodes: every edge pixel that is not a "useless hub" (…_remove_useless_hubs, :294-345)
  std::vector<uint32_t> id_of((size_t)total, 0);
  std::vector<float> cx, cy;
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++)
      if (edge(i, j)) {
        if ((i > 1 && j > 1 && edge(i - 1, j) && edge(i, j - 1) && !edge(i + 1, j + 1)) ||
            (i > 1 && j < cols - 1 && edge(i - 1, j) && edge(i, j + 1) && !edge(i + 1, j - 1)) ||
            (i < rows - 1 && j < cols - 1 && edge(i + 1, j) && edge(i, j + 1) && !edge(i - 1, j - 1)) ||
            (i < rows - 1 && j > 1 && edge(i + 1, j) && edge(i, j - 1) && !edge(i - 1, j + 1))) {
          mask[(size_t)i * cols + j] = 0;
        } else {
          id_of[(size_t)i * cols + j] = (uint32_t)cx.size();
          cx.push_back((float)(j + 0.5));
          cy.push_back((float)(i + 0.5));
        }
      }
  const uint32_t n_px = (uint32_t)cx.size();
  // ---- edges: right, down, down-right, down-left neighbours unless already within 9 steps (:349-426)
  PixelGraph pg(n_px);
  auto link = [&](uint32_t p, int y, int x) {
    if (mask[(size_t)y * cols + x]) {
      const uint32_t c = id_of[(size_t)y * cols + x];
      if (p != c && !pg.reachable(p, c, 8)) pg.add_edge(p, c);
    }
  };
  for (int i = 0; i < rows - 1; i++)
    for (int j = 0; j < cols - 1; j++)
      if (mask[(size_t)i * cols + j]) {
        const uint32_t p = id_of[(size_t)i * cols + j];
        link(p, i, j + 1);
        link(p, i + 1, j);
        link(p, i + 1, j + 1);
        if (j > 1) link(p, i + 1, j - 1);
      }
  // ---- polylines: maximal runs of degree-2 pixels between ends / hubs, in pixel-id order (:428-626)
  Builder B;
  Graph& g = B.g;
  g.init_table(n_px / 4 + 64);
  std::vector<uint8_t> processed(n_px, 0);
  std::vector<uint32_t> run;
  std::vector<float> coords;
  auto deg = [&](uint32_t n) { return (int)pg.deg[n]; };
  auto nb = [&](uint32_t n, int k) { return pg.adj[(size_t)n * 8 + k]; };
  auto walk = [&](uint32_t start, uint32_t no_come_back) {  // find_polylineend_no_come_back
    uint32_t prev = no_come_back, cur = start;
    run.push_back(cur);
    while (cur != no_come_back && deg(cur) == 2) {
      const uint32_t a = nb(cur, 0), b = nb(cur, 1);
      const uint32_t nx = a != prev ? a : b;
      prev = cur;
      cur = nx;
      run.push_back(cur);
    }
  };
  auto emit = [&]() {
    const uint32_t s = run.front(), e = run.back();
    g.node_of(cx[s], cy[s]);
    g.node_of(cx[e], cy[e]);
    if (!(deg(s) > 2)) processed[s] = 1;
    if (!(deg(e) > 2)) processed[e] = 1;
    for (size_t k = 1; k + 1 < run.size(); k++) processed[run[k]] = 1;
    coords.clear();
    for (uint32_t id : run) {
      coords.push_back(cx[id]);
      coords.push_back(cy[id]);
    }
    g.add_run(coords.data(), (uint32_t)run.size());
  };
  for (uint32_t i = 0; i < n_px; i++)
    if (!processed[i]) {
      const int d = deg(i);
      if (d == 2) {
        run.clear();
        walk(nb(i, 0), i);
        std::reverse(run.begin(), run.end());
        run.push_back(i);
        if (run.front() != run.back()) walk(nb(i, 1), i);
        emit();
      } else if (d == 1) {
        run.clear();
        run.push_back(i);
        walk(nb(i, 0), i);
        emit();
      } else if (d > 2) {
        for (int k = 0; k < d; k++) {
          run.clear();
          run.push_back(i);
          walk(nb(i, k), i);
          emit();
        }
      }
      processed[i] = 1;
    }
  B.optimize();
  // ---- flatten
  const size_t NP = g.pls.size(), NN = g.nx.size();
  out->n_polylines = (uint32_t)NP;
  out->n_nodes = (uint32_t)NN;
  out->pl_vtx_off = (uint32_t*)malloc(sizeof(uint32_t) * (NP + 1));
  out->pl_start = (uint32_t*)malloc(sizeof(uint32_t) * (NP + 1));
  out->pl_end = (uint32_t*)malloc(sizeof(uint32_t) * (NP + 1));
  out->pl_valid = (uint8_t*)malloc(NP + 1);
  size_t nv = 0;
  for (const Poly& p : g.pls) nv += p.n;
  out->vtx_xy = (float*)malloc(sizeof(float) * 2 * (nv + 1));
  out->node_xy = (float*)malloc(sizeof(float) * 2 * (NN + 1));
  size_t w = 0;
  for (size_t i = 0; i < NP; i++) {
    const Poly& p = g.pls[i];
    out->pl_vtx_off[i] = (uint32_t)w;
    out->pl_start[i] = p.start;
    out->pl_end[i] = p.end;
    out->pl_valid[i] = g.poly_valid((uint32_t)i) ? 1 : 0;
    if (p.n) memcpy(out->vtx_xy + 2 * w, g.vtx(p), sizeof(float) * 2 * p.n);
    w += p.n;
  }
  out->pl_vtx_off[NP] = (uint32_t)w;
  for (size_t n = 0; n < NN; n++) {
    out->node_xy[2 * n] = g.nx[n];
    out->node_xy[2 * n + 1] = g.ny[n];
  }
  return 0;
}

extern "C" void eg3d_plg_view_free(eg3d_plg_view* v) {
  if (!v) return;
  free(v->pl_vtx_off);
  free(v->vtx_xy);
  free(v->pl_start);
  free(v->pl_end);
  free(v->pl_valid);
  free(v->node_xy);
  memset(v, 0, sizeof(*v));
}

// All views of a scene at once: the views are independent (the reference's loop, convert_edge_images_pixel_to_segment.cpp:
// 868-892, builds them one after the other: 25 dtu006-sized edge maps x ~125 ms = 3 s, sixty times the GPU's share of
// edge_matching()); here they are built on the host's cores. Returns 0, or -(v + 1) for the first view (in order) whose
// image cannot be read or whose size differs from view 0's; on failure nothing is left allocated.
extern "C" int eg3d_plg_build_views_from_png(const char* const* paths, int n_views, int* width, int* height, eg3d_plg_view* out) {
  if (!paths || n_views < 0 || !width || !height || (n_views && !out)) return -1;
  std::vector<int> rc((size_t)n_views, 0), w((size_t)n_views, 0), h((size_t)n_views, 0);
  for (int v = 0; v < n_views; v++) memset(&out[v], 0, sizeof(out[v]));
#pragma omp parallel for schedule(dynamic, 1)
  for (int v = 0; v < n_views; v++) rc[(size_t)v] = eg3d_plg_build_from_png(paths[v], &w[(size_t)v], &h[(size_t)v], &out[v]);
  int bad = 0;
  for (int v = 0; v < n_views && !bad; v++)
    if (rc[(size_t)v] != 0 || w[(size_t)v] != w[0] || h[(size_t)v] != h[0]) bad = -(v + 1);
  if (bad) {
    for (int v = 0; v < n_views; v++)
      if (rc[(size_t)v] == 0) eg3d_plg_view_free(&out[v]);
    return bad;
  }
  *width = n_views ? w[0] : 0;
  *height = n_views ? h[0] : 0;
  return 0;
}

