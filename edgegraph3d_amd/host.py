"""Python view of libeg3d_host.so: synthetic workloads, grid builder, host post steps."""
import ctypes as C
import os

import numpy as np

from . import _cdefs as D

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libeg3d_host.so")
        if not os.path.exists(path):
            raise RuntimeError("libeg3d_host.so is not built: run `python -m edgegraph3d_amd.build`")
        L = C.CDLL(path)
        L.eg3d_synth_default_config.argtypes = [C.POINTER(D.SynthConfig), C.c_int]
        L.eg3d_synth_create.argtypes = [C.POINTER(D.SynthConfig)]
        L.eg3d_synth_create.restype = C.c_void_p
        L.eg3d_synth_scene.argtypes = [C.c_void_p]
        L.eg3d_synth_scene.restype = C.POINTER(D.Scene)
        L.eg3d_synth_seeds.argtypes = [C.c_void_p]
        L.eg3d_synth_seeds.restype = C.POINTER(D.Seeds)
        L.eg3d_synth_seed_truth.argtypes = [C.c_void_p]
        L.eg3d_synth_seed_truth.restype = D.f32p
        L.eg3d_synth_total_segments.argtypes = [C.c_void_p]
        L.eg3d_synth_total_segments.restype = C.c_uint64
        L.eg3d_synth_destroy.argtypes = [C.c_void_p]
        L.eg3d_synth_polyline_curve.argtypes = [C.c_void_p]
        L.eg3d_synth_polyline_curve.restype = D.u32p
        L.eg3d_synth_n_curves.argtypes = [C.c_void_p]
        L.eg3d_synth_points.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(D.f32p), C.POINTER(D.u32p),
                                        C.POINTER(D.i32p), C.POINTER(D.f32p)]
        L.eg3d_host_free.argtypes = [C.c_void_p]
        L.eg3d_host_build_grid.argtypes = [C.POINTER(D.Scene), C.c_int, C.c_float, D.u32p, D.u32p,
                                           C.POINTER(D.u32p), C.POINTER(D.u32p), D.u32p]
        L.eg3d_host_filter_close_2d.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(D.EdgePoints), D.u8p]
        L.eg3d_host_observation_filter.argtypes = [C.c_int, D.u32p, C.c_uint64, C.c_uint64, C.c_int, D.u8p]
        L.eg3d_host_replay_matches.argtypes = [C.POINTER(D.Scene), C.POINTER(D.EdgePoints), C.POINTER(D.Graph3D)]
        L.eg3d_host_free_graph3d.argtypes = [C.POINTER(D.Graph3D)]
        L.eg3d_plg_build_from_mask.argtypes = [D.u8p, C.c_int, C.c_int, C.POINTER(D.PlgView)]
        L.eg3d_plg_build_from_png.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(D.PlgView)]
        L.eg3d_png_read_edge_mask.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(D.u8p)]
        L.eg3d_plg_view_free.argtypes = [C.POINTER(D.PlgView)]
        L.eg3d_plg_from_views.restype = C.c_void_p
        L.eg3d_plg_from_views.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(D.PlgView)]
        L.eg3d_plg_scene.restype = C.POINTER(D.Scene)
        L.eg3d_plg_scene.argtypes = [C.c_void_p]
        L.eg3d_plg_destroy.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def default_config(index):
    c = D.SynthConfig()
    lib().eg3d_synth_default_config(C.byref(c), index)
    return c


class Synth:
    """Seeded synthetic scene + seeds (owned by the native library)."""

    def __init__(self, config):
        if isinstance(config, int):
            config = default_config(config)
        self.config = config
        self._h = lib().eg3d_synth_create(C.byref(config))
        if not self._h:
            raise RuntimeError("eg3d_synth_create failed")
        self.scene = lib().eg3d_synth_scene(self._h)   # POINTER(Scene)
        self.seeds = lib().eg3d_synth_seeds(self._h)   # POINTER(Seeds)

    def close(self):
        if self._h:
            lib().eg3d_synth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_views(self):
        return int(self.scene.contents.n_views)

    @property
    def n_seeds(self):
        return int(self.seeds.contents.n_seeds)

    @property
    def total_segments(self):
        return int(lib().eg3d_synth_total_segments(self._h))

    def seeds_np(self):
        s = self.seeds.contents
        n = int(s.n_seeds)
        off = D.as_np(s.trk_off, n + 1, np.uint32)
        m = int(off[-1])
        return off, D.as_np(s.trk_view, m, np.int32), D.as_np(s.trk_xy, 2 * m, np.float32).reshape(m, 2)

    def seed_truth(self):
        """[n_seeds, 3] noise-free 3-D positions the seeds were generated from"""
        return D.as_np(lib().eg3d_synth_seed_truth(self._h), 3 * self.n_seeds, np.float32).reshape(-1, 3).astype(np.float64)

    def scene_np(self):
        s = self.scene.contents
        V = int(s.n_views)
        vpo = D.as_np(s.view_pl_off, V + 1, np.uint32)
        NP = int(vpo[-1])
        pvo = D.as_np(s.pl_vtx_off, NP + 1, np.uint32)
        NV = int(pvo[-1])
        return {
            "n_views": V, "width": int(s.width), "height": int(s.height),
            "cam_P": D.as_np(s.cam_P, V * 16, np.float32).reshape(V, 16),
            "F": D.as_np(s.F, V * V * 9, np.float64).reshape(V, V, 9),
            "F_valid": D.as_np(s.F_valid, V * V, np.uint8).reshape(V, V),
            "view_pl_off": vpo, "pl_vtx_off": pvo,
            "vtx_xy": D.as_np(s.vtx_xy, 2 * NV, np.float32).reshape(NV, 2),
            "pl_start": D.as_np(s.pl_start, NP, np.uint32), "pl_end": D.as_np(s.pl_end, NP, np.uint32),
            "pl_valid": D.as_np(s.pl_valid, NP, np.uint8),
        }

    def polyline_sets(self, max_sets=None):
        """Synthetic "potentially compatible polylines" sets (the input of pipelines 1-2, produced in
        the reference by the polyline matcher): one set per 3-D curve = the valid polylines of every
        view generated from it. Returns (n_sets, row_off[n_sets*V+1], pl_ids) — a CSR over rows
        (set * V + view), view-local polyline ids ascending per row."""
        sc = self.scene_np()
        V, vpo = sc["n_views"], sc["view_pl_off"]
        NP = int(vpo[-1])
        curve = D.as_np(lib().eg3d_synth_polyline_curve(self._h), NP, np.uint32)
        n_sets = int(lib().eg3d_synth_n_curves(self._h))
        if max_sets is not None:
            n_sets = min(n_sets, max_sets)
        nvtx = np.diff(sc["pl_vtx_off"])
        row_off, ids = [0], []
        for c in range(n_sets):
            for v in range(V):
                a, b = int(vpo[v]), int(vpo[v + 1])
                sel = np.nonzero((curve[a:b] == c) & (sc["pl_valid"][a:b] != 0) & (nvtx[a:b] >= 2))[0]
                ids.extend(int(i) for i in sel)
                row_off.append(len(ids))
        return n_sets, np.asarray(row_off, np.uint32), np.asarray(ids if ids else [0], np.uint32)[:len(ids)]

    def points(self, n_points, rng_seed=0xC5):
        """Config-5 workload: returns X, obs_off, obs_view, obs_xy (numpy copies)."""
        X, off, view, xy = D.f32p(), D.u32p(), D.i32p(), D.f32p()
        rc = lib().eg3d_synth_points(self._h, n_points, rng_seed, C.byref(X), C.byref(off), C.byref(view), C.byref(xy))
        if rc != 0:
            raise RuntimeError("eg3d_synth_points failed")
        o = D.as_np(off, n_points + 1, np.uint32)
        m = int(o[-1])
        out = (D.as_np(X, 3 * n_points, np.float32).reshape(n_points, 3), o, D.as_np(view, m, np.int32),
               D.as_np(xy, 2 * m, np.float32).reshape(m, 2))
        for p in (X, off, view, xy):
            lib().eg3d_host_free(p)
        return out


def replay_matches(scene_ptr, cloud):
    """Row a17: the 3-D polyline graph and matched 2-D intervals PLGMatchesManager would hold after
    the path emitted `cloud` (a dict as returned by Context.match_refpoints)."""
    ep = D.EdgePointsArrays(cloud)
    g = D.Graph3D()
    rc = lib().eg3d_host_replay_matches(scene_ptr, C.byref(ep.c), C.byref(g))
    if rc != 0:
        raise RuntimeError("eg3d_host_replay_matches failed (%d)" % rc)
    d = D.graph3d_to_dict(g)
    lib().eg3d_host_free_graph3d(C.byref(g))
    return d


def plg_from_mask(mask):
    """binary edge image (uint8 [h, w], non-zero = edge) -> optimised polyline graph of one view."""
    m = np.ascontiguousarray(mask, np.uint8)
    v = D.PlgView()
    rc = lib().eg3d_plg_build_from_mask(D.np_ptr(m, C.c_uint8), m.shape[1], m.shape[0], C.byref(v))
    if rc != 0:
        raise RuntimeError("eg3d_plg_build_from_mask failed (%d)" % rc)
    d = D.plg_view_to_dict(v)
    lib().eg3d_plg_view_free(C.byref(v))
    return d


def estimate_F(n_views, trk_off, trk_view, trk_xy, estimate=True, rng_seed=0):
    """SURVEY N4 (geometric_utilities.cpp:754-820): fundamental matrices of all ordered view pairs from the
    point tracks; pairs with fewer than 10 common points are invalid. Returns (F [V,V,9] f64, valid [V,V] u8,
    n_common [V,V] u32, pairs whose estimate failed). estimate=False: validity rule and counts only."""
    L = lib()
    L.eg3d_host_estimate_F.restype = C.c_int
    L.eg3d_host_estimate_F.argtypes = [C.c_int, C.c_uint64, D.u32p, D.i32p, D.f32p, C.c_int, C.c_uint64, D.f64p,
                                       D.u8p, D.u32p]
    off = np.ascontiguousarray(trk_off, np.uint32)
    view = np.ascontiguousarray(trk_view, np.int32)
    xy = np.ascontiguousarray(trk_xy, np.float32)
    V = int(n_views)
    F = np.zeros((V, V, 9), np.float64)
    valid = np.zeros((V, V), np.uint8)
    ncom = np.zeros((V, V), np.uint32)
    rc = L.eg3d_host_estimate_F(V, len(off) - 1, D.np_ptr(off, C.c_uint32), D.np_ptr(view, C.c_int32),
                                D.np_ptr(xy, C.c_float), 1 if estimate else 0, rng_seed, D.np_ptr(F, C.c_double),
                                D.np_ptr(valid, C.c_uint8), D.np_ptr(ncom, C.c_uint32))
    if rc < 0:
        raise RuntimeError("eg3d_host_estimate_F failed (%d)" % rc)
    return F, valid, ncom, rc


def png_edge_mask(path):
    w, h, p = C.c_int(), C.c_int(), D.u8p()
    rc = lib().eg3d_png_read_edge_mask(path.encode(), C.byref(w), C.byref(h), C.byref(p))
    if rc != 0:
        raise RuntimeError("eg3d_png_read_edge_mask(%s) failed (%d)" % (path, rc))
    m = D.as_np(p, w.value * h.value, np.uint8).reshape(h.value, w.value)
    lib().eg3d_host_free(p)
    return m


def build_grid(scene_ptr, view, cell_dim):
    ncols, nrows, dropped = C.c_uint32(), C.c_uint32(), C.c_uint32()
    off, ids = D.u32p(), D.u32p()
    rc = lib().eg3d_host_build_grid(scene_ptr, view, cell_dim, C.byref(ncols), C.byref(nrows), C.byref(off),
                                    C.byref(ids), C.byref(dropped))
    if rc != 0:
        raise RuntimeError("eg3d_host_build_grid failed")
    n = ncols.value * nrows.value
    o = D.as_np(off, n + 1, np.uint32)
    i = D.as_np(ids, int(o[-1]), np.uint32)
    lib().eg3d_host_free(off)
    lib().eg3d_host_free(ids)
    return ncols.value, nrows.value, o, i, dropped.value


class SeedsArrays:
    """Owns numpy copies of a seed set and exposes a ctypes Seeds struct over them."""

    def __init__(self, trk_off, trk_view, trk_xy):
        self.trk_off = np.ascontiguousarray(trk_off, dtype=np.uint32)
        self.trk_view = np.ascontiguousarray(trk_view, dtype=np.int32)
        self.trk_xy = np.ascontiguousarray(trk_xy, dtype=np.float32).reshape(-1, 2)
        self.c = D.Seeds(len(self.trk_off) - 1, D.np_ptr(self.trk_off, C.c_uint32), D.np_ptr(self.trk_view, C.c_int32),
                         D.np_ptr(self.trk_xy, C.c_float))


class SceneArrays:
    """Owns numpy arrays of a scene and exposes a ctypes Scene struct over them."""

    def __init__(self, d):
        self.d = {
            "cam_P": np.ascontiguousarray(d["cam_P"], np.float32), "F": np.ascontiguousarray(d["F"], np.float64),
            "F_valid": np.ascontiguousarray(d["F_valid"], np.uint8),
            "view_pl_off": np.ascontiguousarray(d["view_pl_off"], np.uint32),
            "pl_vtx_off": np.ascontiguousarray(d["pl_vtx_off"], np.uint32),
            "vtx_xy": np.ascontiguousarray(d["vtx_xy"], np.float32),
            "pl_start": np.ascontiguousarray(d["pl_start"], np.uint32),
            "pl_end": np.ascontiguousarray(d["pl_end"], np.uint32),
            "pl_valid": np.ascontiguousarray(d["pl_valid"], np.uint8),
        }
        a = self.d
        self.c = D.Scene(int(d["n_views"]), int(d["width"]), int(d["height"]), D.np_ptr(a["cam_P"], C.c_float),
                         D.np_ptr(a["F"], C.c_double), D.np_ptr(a["F_valid"], C.c_uint8),
                         D.np_ptr(a["view_pl_off"], C.c_uint32), D.np_ptr(a["pl_vtx_off"], C.c_uint32),
                         D.np_ptr(a["vtx_xy"], C.c_float), D.np_ptr(a["pl_start"], C.c_uint32),
                         D.np_ptr(a["pl_end"], C.c_uint32), D.np_ptr(a["pl_valid"], C.c_uint8))
