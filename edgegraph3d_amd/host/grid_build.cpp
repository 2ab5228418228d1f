// Uniform-grid construction for one view, host side of eg3d_create.
//
// Behaviour reproduced: PolyLine2DMap ctor + polyline::get_intersectedcells_2dmap_set
// (reference: src/edgegraph3d/matching/plg_matching/polyLine_2d_map.cpp:40-58,
//  src/edgegraph3d/plgs/polyline_graph_2d.cpp:555-577,819-835): every VALID polyline is
// sampled from its start every cell/(1.414+0.1) px (Euclidean stepping), samples lying on a
// cell boundary are dropped, and each remaining sample's cell lists the polyline once.
// Own design: (cell, polyline) pairs are generated per polyline into a flat vector and turned
// into a CSR by a counting sort, instead of a 2-D array of std::vector.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/eg3d_host.h"
#include "eg3d_dev_geom.h"

using namespace eg3d;

extern "C" int eg3d_host_build_grid(const eg3d_scene* sc, int view, float cell_dim, uint32_t* ncols, uint32_t* nrows,
                                    uint32_t** cell_off_out, uint32_t** ids_out, uint32_t* dropped_out) {
  if (!sc || view < 0 || view >= sc->n_views || !(cell_dim > 0)) return -1;
  const int map_w = (int)std::ceil(sc->width / cell_dim);
  const int map_h = (int)std::ceil(sc->height / cell_dim);
  const size_t n_cells = (size_t)map_w * map_h;
  const float step = (float)(cell_dim / (1.414 + 0.1));
  std::vector<uint32_t> pair_cell, pair_pl;
  uint32_t dropped = 0;
  const uint32_t p0 = sc->view_pl_off[view], p1 = sc->view_pl_off[view + 1];
  std::vector<uint32_t> local;  // cells of the current polyline
  for (uint32_t g = p0; g < p1; g++) {
    if (!sc->pl_valid[g]) continue;
    PlRef pl;
    pl.v = reinterpret_cast<const f2*>(sc->vtx_xy) + sc->pl_vtx_off[g];
    pl.n = sc->pl_vtx_off[g + 1] - sc->pl_vtx_off[g];
    pl.start = sc->pl_start[g];
    pl.end = sc->pl_end[g];
    if (pl.n < 2) continue;
    local.clear();
    // walk from `start` towards the other extreme; for a loop polyline (start == end) the
    // "other end" is start itself and the walk stops at once (Q8)
    const uint32_t direction = pl.end;  // get_other_end(start): end, which equals start for loops
    PlPt cur;
    cur.seg = 0;
    cur.x = pl.v[0].x;
    cur.y = pl.v[0].y;
    bool have_prev = false;
    int32_t prev_c = 0, prev_r = 0;
    auto visit = [&](const PlPt& p) {
      CellCoord cc = cell_of(cell_dim, p.x, p.y);
      if (cc.bx || cc.by) return;
      if (have_prev && local.size() > 0 && cc.col == prev_c && cc.row == prev_r) return;
      if (cc.col < 0 || cc.col >= map_w || cc.row < 0 || cc.row >= map_h) {
        dropped++;  // the reference indexes out of bounds here; inputs must keep vertices inside the image
      } else {
        local.push_back((uint32_t)(cc.row * map_w + cc.col));
      }
      prev_c = cc.col;
      prev_r = cc.row;
      have_prev = true;
    };
    visit(cur);
    for (;;) {
      PlPt nx;
      uint32_t w = walk_by_distance(pl, cur, direction, step, nx);
      visit(nx);
      cur = nx;
      if (w & WALK_EXTREME) break;
    }
    std::sort(local.begin(), local.end());
    local.erase(std::unique(local.begin(), local.end()), local.end());
    for (uint32_t c : local) {
      pair_cell.push_back(c);
      pair_pl.push_back(g - p0);
    }
  }
  uint32_t* off = (uint32_t*)malloc(sizeof(uint32_t) * (n_cells + 1));
  uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * (pair_cell.size() ? pair_cell.size() : 1));
  memset(off, 0, sizeof(uint32_t) * (n_cells + 1));
  for (uint32_t c : pair_cell) off[c + 1]++;
  for (size_t c = 0; c < n_cells; c++) off[c + 1] += off[c];
  std::vector<uint32_t> fill(off, off + n_cells);
  for (size_t i = 0; i < pair_cell.size(); i++) ids[fill[pair_cell[i]]++] = pair_pl[i];  // stable: ids ascending
  *ncols = (uint32_t)map_w;
  *nrows = (uint32_t)map_h;
  *cell_off_out = off;
  *ids_out = ids;
  if (dropped_out) *dropped_out = dropped;
  return 0;
}
