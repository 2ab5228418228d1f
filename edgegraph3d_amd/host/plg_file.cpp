// Polyline-graph container file. The reference builds its PolyLineGraph2DHMapImpl per view from
// the edge images at start-up (convert_edge_images_pixel_to_segment.cpp, SURVEY N2 — rebuilt in
// plg_build.cpp); this small binary container lets any producer of polyline graphs (that builder, or
// the reference's own) hand them to the path:
//   "EG3DPLG1"  i32 n_views  i32 width  i32 height
//   per view:   u32 n_polylines, then per polyline: u32 start_node  u32 end_node  u8 valid  u32 n_vtx  f32 xy[2*n_vtx]
// Little endian. Polyline ids are positions in the file (the reference's vector index).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/eg3d_host.h"

struct eg3d_plg {
  eg3d_scene scene;
  std::vector<uint32_t> view_pl_off, pl_vtx_off, pl_start, pl_end;
  std::vector<uint8_t> pl_valid;
  std::vector<float> vtx;
};

extern "C" int eg3d_plg_write(const char* path, const eg3d_scene* sc) {
  if (!path || !sc) return -1;
  FILE* f = fopen(path, "wb");
  if (!f) return -2;
  fwrite("EG3DPLG1", 1, 8, f);
  int32_t hdr[3] = {sc->n_views, sc->width, sc->height};
  fwrite(hdr, 4, 3, f);
  for (int v = 0; v < sc->n_views; v++) {
    const uint32_t a = sc->view_pl_off[v], b = sc->view_pl_off[v + 1];
    const uint32_t n = b - a;
    fwrite(&n, 4, 1, f);
    for (uint32_t p = a; p < b; p++) {
      const uint32_t nv = sc->pl_vtx_off[p + 1] - sc->pl_vtx_off[p];
      fwrite(&sc->pl_start[p], 4, 1, f);
      fwrite(&sc->pl_end[p], 4, 1, f);
      fwrite(&sc->pl_valid[p], 1, 1, f);
      fwrite(&nv, 4, 1, f);
      fwrite(sc->vtx_xy + 2 * (size_t)sc->pl_vtx_off[p], 4, 2 * (size_t)nv, f);
    }
  }
  return fclose(f) == 0 ? 0 : -3;
}

extern "C" eg3d_plg* eg3d_plg_read(const char* path) {
  FILE* f = path ? fopen(path, "rb") : nullptr;
  if (!f) return nullptr;
  char magic[8];
  int32_t hdr[3];
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "EG3DPLG1", 8) != 0 || fread(hdr, 4, 3, f) != 3 || hdr[0] < 1) {
    fclose(f);
    return nullptr;
  }
  eg3d_plg* g = new eg3d_plg();
  g->view_pl_off.push_back(0);
  g->pl_vtx_off.push_back(0);
  bool ok = true;
  for (int v = 0; v < hdr[0] && ok; v++) {
    uint32_t n = 0;
    ok = fread(&n, 4, 1, f) == 1;
    for (uint32_t p = 0; p < n && ok; p++) {
      uint32_t st, en, nv;
      uint8_t valid;
      ok = fread(&st, 4, 1, f) == 1 && fread(&en, 4, 1, f) == 1 && fread(&valid, 1, 1, f) == 1 && fread(&nv, 4, 1, f) == 1 &&
           nv < (1u << 28);
      if (!ok) break;
      const size_t o = g->vtx.size();
      g->vtx.resize(o + 2 * (size_t)nv);
      ok = nv == 0 || fread(g->vtx.data() + o, 4, 2 * (size_t)nv, f) == 2 * (size_t)nv;
      g->pl_start.push_back(st);
      g->pl_end.push_back(en);
      g->pl_valid.push_back(valid);
      g->pl_vtx_off.push_back((uint32_t)(g->vtx.size() / 2));
    }
    g->view_pl_off.push_back((uint32_t)g->pl_start.size());
  }
  fclose(f);
  if (!ok) {
    delete g;
    return nullptr;
  }
  memset(&g->scene, 0, sizeof(g->scene));
  g->scene.n_views = hdr[0];
  g->scene.width = hdr[1];
  g->scene.height = hdr[2];
  g->scene.view_pl_off = g->view_pl_off.data();
  g->scene.pl_vtx_off = g->pl_vtx_off.data();
  g->scene.vtx_xy = g->vtx.data();
  g->scene.pl_start = g->pl_start.data();
  g->scene.pl_end = g->pl_end.data();
  g->scene.pl_valid = g->pl_valid.data();
  return g;
}

// per-view graphs of the N2 builder (eg3d_plg_build_from_mask / _png) -> the container the path consumes
extern "C" eg3d_plg* eg3d_plg_from_views(int n_views, int width, int height, const eg3d_plg_view* views) {
  if (n_views < 1 || !views) return nullptr;
  eg3d_plg* g = new eg3d_plg();
  g->view_pl_off.push_back(0);
  g->pl_vtx_off.push_back(0);
  for (int v = 0; v < n_views; v++) {
    const eg3d_plg_view& pv = views[v];
    for (uint32_t p = 0; p < pv.n_polylines; p++) {
      const uint32_t a = pv.pl_vtx_off[p], b = pv.pl_vtx_off[p + 1];
      g->vtx.insert(g->vtx.end(), pv.vtx_xy + 2 * (size_t)a, pv.vtx_xy + 2 * (size_t)b);
      g->pl_start.push_back(pv.pl_start[p]);
      g->pl_end.push_back(pv.pl_end[p]);
      g->pl_valid.push_back(pv.pl_valid[p]);
      g->pl_vtx_off.push_back((uint32_t)(g->vtx.size() / 2));
    }
    g->view_pl_off.push_back((uint32_t)g->pl_start.size());
  }
  if (g->vtx.empty()) g->vtx.assign(2, 0.f);
  memset(&g->scene, 0, sizeof(g->scene));
  g->scene.n_views = n_views;
  g->scene.width = width;
  g->scene.height = height;
  g->scene.view_pl_off = g->view_pl_off.data();
  g->scene.pl_vtx_off = g->pl_vtx_off.data();
  g->scene.vtx_xy = g->vtx.data();
  g->scene.pl_start = g->pl_start.data();
  g->scene.pl_end = g->pl_end.data();
  g->scene.pl_valid = g->pl_valid.data();
  return g;
}

// the polyline part of an eg3d_scene (cam_P, F, F_valid are null: the caller fills them in)
extern "C" const eg3d_scene* eg3d_plg_scene(const eg3d_plg* g) { return g ? &g->scene : nullptr; }
extern "C" void eg3d_plg_destroy(eg3d_plg* g) { delete g; }
