"""ctypes binding of the CPU oracle (oracle/liboracle.so).

ORACLE = TEST INFRASTRUCTURE ONLY (parity unpinned, see oracle/eg3d_oracle.cpp). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package edgegraph3d_amd never does.
"""
import ctypes as C
import os

import numpy as np

from edgegraph3d_amd import _cdefs as D

_LIB = None


class Stats(C.Structure):
    _fields_ = [("n_tasks", C.c_uint64), ("n_hyp", C.c_uint64), ("n_chains", C.c_uint64), ("n_tri", C.c_uint64),
                ("n_add", C.c_uint64), ("n_degenerate_dlt", C.c_uint64), ("n_combos", C.c_uint64),
                ("bytes_algorithmic", C.c_uint64), ("dir_mismatch", C.c_uint32), ("grid_dropped", C.c_uint32),
                ("seconds", C.c_double)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so is not built: run `make -C oracle`")
        L = C.CDLL(path)
        L.orc_create.argtypes = [C.POINTER(D.Scene)]
        L.orc_create.restype = C.c_void_p
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_get_grid.argtypes = [C.c_void_p, C.c_int, C.c_int, D.u32p, D.u32p, C.POINTER(D.u32p), C.POINTER(D.u32p)]
        L.orc_match_refpoints.argtypes = [C.c_void_p, C.POINTER(D.Seeds), C.c_uint32, C.c_uint32, C.c_int,
                                          C.POINTER(D.EdgePoints), C.POINTER(Stats)]
        L.orc_match_polyline_sets.argtypes = [C.c_void_p, C.c_uint32, D.u32p, D.u32p, C.c_uint32, C.c_uint32, C.c_int,
                                              C.POINTER(D.EdgePoints), C.POINTER(Stats)]
        L.orc_free_edgepoints.argtypes = [C.POINTER(D.EdgePoints)]
        L.orc_candidates.argtypes = [C.c_void_p, C.POINTER(D.Seeds), C.c_uint32, C.c_uint32, C.POINTER(D.Candidates)]
        L.orc_free_candidates.argtypes = [C.POINTER(D.Candidates)]
        L.orc_gn_filter.argtypes = [C.c_void_p, D.f32p, D.u32p, D.i32p, D.f32p, C.c_uint64, C.c_float, C.c_int,
                                    C.c_int, D.f32p, D.u8p]
        L.orc_filter_close_2d.argtypes = [C.c_void_p, C.POINTER(D.EdgePoints), D.u8p]
        L.orc_observation_filter.argtypes = [C.c_int, D.u32p, C.c_uint64, C.c_uint64, C.c_int, D.u8p]
        L.orc_replay_matches.argtypes = [C.c_void_p, C.POINTER(D.EdgePoints), C.POINTER(D.Graph3D)]
        L.orc_free_graph3d.argtypes = [C.POINTER(D.Graph3D)]
        L.orc_plg_from_mask.argtypes = [D.u8p, C.c_int, C.c_int, C.POINTER(D.PlgView)]
        L.orc_free_plg_view.argtypes = [C.POINTER(D.PlgView)]
        L.orc_squared_2d_distance.restype = C.c_float
        L.orc_squared_2d_distance.argtypes = [C.c_float] * 4
        L.orc_minimum_distancesq.restype = C.c_float
        L.orc_minimum_distancesq.argtypes = [C.c_float] * 6 + [D.f32p]
        L.orc_intersect_segment_line.argtypes = [C.c_float] * 4 + [D.f32p, D.f32p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_intersect_segment_line_nqp.argtypes = [C.c_float] * 4 + [D.f32p, D.f32p, C.POINTER(C.c_int), D.f32p]
        L.orc_cell_from_coords.argtypes = [C.c_float] * 3 + [C.POINTER(C.c_int)] * 4
        L.orc_epiline.argtypes = [D.f64p, C.c_float, C.c_float, D.f32p]
        L.orc_project.argtypes = [D.f32p, D.f32p, D.f32p]
        L.orc_triangulate.argtypes = [D.f32p, C.POINTER(C.c_int), D.f32p, C.c_int, D.f32p, C.POINTER(C.c_int)]
        L.orc_gn_add.argtypes = [D.f32p, C.POINTER(C.c_int), D.f32p, C.c_int, D.f32p, D.f32p]
        L.orc_dlt.argtypes = [D.f32p, D.f32p, D.f32p, D.f32p, D.f64p]
        L.orc_next_by_distance.argtypes = [D.f32p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                           C.c_uint32, C.c_float, D.u32p, D.f32p]
        L.orc_next_by_line.argtypes = [D.f32p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float,
                                       C.c_uint32, D.f32p, C.c_int, C.c_float, C.c_float, D.u32p, D.f32p,
                                       C.POINTER(C.c_int)]
        # the checker follows the DLT form the product library was built with (include/eg3d.h eg3d_dlt_rows)
        rows = os.environ.get("EG3D_ORACLE_DLT_ROWS")
        if rows is None:
            try:
                from edgegraph3d_amd import api
                rows = api.lib().eg3d_dlt_rows()
            except Exception:
                rows = 3
        assert L.orc_set_dlt_rows(int(rows)) == 0
        _LIB = L
    return _LIB


def plg_from_mask(mask):
    """SURVEY N2 (convert_edge_images_pixel_to_segment.cpp:879-883): edge mask -> optimised polyline graph."""
    m = np.ascontiguousarray(mask, np.uint8)
    v = D.PlgView()
    assert lib().orc_plg_from_mask(D.np_ptr(m, C.c_uint8), m.shape[1], m.shape[0], C.byref(v)) == 0
    d = D.plg_view_to_dict(v)
    lib().orc_free_plg_view(C.byref(v))
    return d


class Oracle:
    def __init__(self, scene_ptr):
        self._h = lib().orc_create(scene_ptr)
        if not self._h:
            raise RuntimeError("orc_create failed")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def grid(self, view, which):
        ncols, nrows = C.c_uint32(), C.c_uint32()
        off, ids = D.u32p(), D.u32p()
        rc = lib().orc_get_grid(self._h, view, which, C.byref(ncols), C.byref(nrows), C.byref(off), C.byref(ids))
        assert rc == 0
        n = ncols.value * nrows.value
        o = D.as_np(off, n + 1, np.uint32)
        return ncols.value, nrows.value, o, D.as_np(ids, int(o[-1]), np.uint32)

    def match(self, seeds_ptr, begin, end, nthreads=1):
        e, st = D.EdgePoints(), Stats()
        rc = lib().orc_match_refpoints(self._h, seeds_ptr, begin, end, nthreads, C.byref(e), C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_match_refpoints failed")
        d = D.edgepoints_to_dict(e)
        lib().orc_free_edgepoints(C.byref(e))
        d["stats"] = {f[0]: getattr(st, f[0]) for f in Stats._fields_}
        return d

    def match_polyline_sets(self, n_sets, row_off, pl_ids, begin=0, end=None, nthreads=1):
        """Pipelines 1-2 extractor on sets [begin, end)."""
        if end is None:
            end = n_sets
        row_off = np.ascontiguousarray(row_off, np.uint32)
        pl_ids = np.ascontiguousarray(pl_ids if len(pl_ids) else [0], np.uint32)
        e, st = D.EdgePoints(), Stats()
        rc = lib().orc_match_polyline_sets(self._h, n_sets, D.np_ptr(row_off, C.c_uint32), D.np_ptr(pl_ids, C.c_uint32),
                                           begin, end, nthreads, C.byref(e), C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_match_polyline_sets failed")
        d = D.edgepoints_to_dict(e)
        lib().orc_free_edgepoints(C.byref(e))
        d["stats"] = {f[0]: getattr(st, f[0]) for f in Stats._fields_}
        return d

    def replay_matches(self, cloud):
        """Row a17 (plg_matches_manager.cpp:99-180) over the chains of `cloud` (an edge-point dict)."""
        ep = D.EdgePointsArrays(cloud)
        g = D.Graph3D()
        assert lib().orc_replay_matches(self._h, C.byref(ep.c), C.byref(g)) == 0
        d = D.graph3d_to_dict(g)
        lib().orc_free_graph3d(C.byref(g))
        return d

    def candidates(self, seeds_ptr, begin, end):
        c = D.Candidates()
        rc = lib().orc_candidates(self._h, seeds_ptr, begin, end, C.byref(c))
        if rc != 0:
            raise RuntimeError("orc_candidates failed")
        d = D.candidates_to_dict(c)
        lib().orc_free_candidates(C.byref(c))
        return d

    def candidates_raw(self, seeds_ptr, begin, end):
        c = D.Candidates()
        rc = lib().orc_candidates(self._h, seeds_ptr, begin, end, C.byref(c))
        if rc != 0:
            raise RuntimeError("orc_candidates failed")
        return c

    def gn_filter(self, X, obs_off, obs_view, obs_xy, gn_max_mse, legacy_abs=False, nthreads=1):
        X = np.ascontiguousarray(X, np.float32)
        obs_off = np.ascontiguousarray(obs_off, np.uint32)
        obs_view = np.ascontiguousarray(obs_view, np.int32)
        obs_xy = np.ascontiguousarray(obs_xy, np.float32)
        n = len(obs_off) - 1
        Xo = np.zeros((n, 3), np.float32)
        inl = np.zeros(n, np.uint8)
        rc = lib().orc_gn_filter(self._h, D.np_ptr(X, C.c_float), D.np_ptr(obs_off, C.c_uint32),
                                 D.np_ptr(obs_view, C.c_int32), D.np_ptr(obs_xy, C.c_float), n, gn_max_mse,
                                 1 if legacy_abs else 0, nthreads, D.np_ptr(Xo, C.c_float), D.np_ptr(inl, C.c_uint8))
        assert rc == 0
        return Xo, inl
