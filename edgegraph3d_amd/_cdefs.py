"""ctypes mirrors of the POD structs in include/eg3d.h and include/eg3d_host.h (plumbing only)."""
import ctypes as C

import numpy as np

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


class Scene(C.Structure):
    _fields_ = [("n_views", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("cam_P", f32p), ("F", f64p), ("F_valid", u8p), ("view_pl_off", u32p),
                ("pl_vtx_off", u32p), ("vtx_xy", f32p), ("pl_start", u32p), ("pl_end", u32p),
                ("pl_valid", u8p)]


class Seeds(C.Structure):
    _fields_ = [("n_seeds", C.c_uint32), ("trk_off", u32p), ("trk_view", i32p), ("trk_xy", f32p)]


class PolylineSets(C.Structure):
    _fields_ = [("n_sets", C.c_uint32), ("row_off", u32p), ("pl_ids", u32p)]


class EdgePoints(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_obs", C.c_uint64), ("X", f32p), ("obs_off", u64p),
                ("obs_view", i32p), ("obs_pl", u32p), ("obs_seg", u32p), ("obs_xy", f32p), ("key", u32p),
                ("n_tasks", C.c_uint64), ("n_hypotheses", C.c_uint64), ("n_chains", C.c_uint64),
                ("flags", C.c_uint32), ("_owner", C.c_void_p)]


class Candidates(C.Structure):
    _fields_ = [("n_sv", C.c_uint32), ("cand_off", u32p), ("cand_pl", u32p), ("start_off", u32p),
                ("start_pl", u32p), ("start_seg", u32p), ("start_xy", f32p), ("n_tasks", C.c_uint32),
                ("task_sv", u32p), ("task_hit", u32p), ("task_list_off", u32p), ("list_off", u32p),
                ("hit_pl", u32p), ("hit_seg", u32p), ("hit_xy", f32p), ("_owner", C.c_void_p)]


class StageTimes(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_candidates", C.c_float), ("ms_epipolar", C.c_float),
                ("ms_hypotheses", C.c_float), ("ms_select", C.c_float), ("ms_expand", C.c_float),
                ("ms_emit", C.c_float), ("bytes_algorithmic", C.c_uint64), ("ms_slowest_chain", C.c_float)]


class DeviceEdgePoints(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_obs", C.c_uint64), ("X", C.c_void_p), ("obs_off", C.c_void_p),
                ("obs_view", C.c_void_p), ("obs_pl", C.c_void_p), ("obs_seg", C.c_void_p), ("obs_xy", C.c_void_p),
                ("key", C.c_void_p), ("complete", C.c_int32)]


class SynthConfig(C.Structure):
    _fields_ = [("n_views", C.c_int32), ("n_seeds", C.c_uint32), ("n_curves", C.c_int32),
                ("rng_seed", C.c_uint64), ("max_track", C.c_int32), ("obs_noise_px", C.c_float),
                ("vtx_noise_px", C.c_float), ("invalid_frac", C.c_float), ("seed_offset_px", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("focal", C.c_float), ("ppx", C.c_float),
                ("ppy", C.c_float)]


def as_np(ptr, n, dtype):
    """Copy n elements behind a ctypes pointer into a numpy array."""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


def np_ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class PlgView(C.Structure):
    """eg3d_plg_view (include/eg3d_host.h): the polyline graph of one view."""
    _fields_ = [("n_polylines", C.c_uint32), ("pl_vtx_off", u32p), ("vtx_xy", f32p), ("pl_start", u32p),
                ("pl_end", u32p), ("pl_valid", u8p), ("n_nodes", C.c_uint32), ("node_xy", f32p)]


def plg_view_to_dict(v):
    n, nn = int(v.n_polylines), int(v.n_nodes)
    off = as_np(v.pl_vtx_off, n + 1, np.uint32)
    nv = int(off[-1]) if n else 0
    return {"n_polylines": n, "n_nodes": nn, "pl_vtx_off": off,
            "vtx_xy": as_np(v.vtx_xy, 2 * nv, np.float32).reshape(nv, 2),
            "pl_start": as_np(v.pl_start, n, np.uint32), "pl_end": as_np(v.pl_end, n, np.uint32),
            "pl_valid": as_np(v.pl_valid, n, np.uint8), "node_xy": as_np(v.node_xy, 2 * nn, np.float32).reshape(nn, 2)}


class Graph3D(C.Structure):
    """eg3d_graph3d (include/eg3d_host.h): the PLGMatchesManager replay (row a17)."""
    _fields_ = [("n_nodes", C.c_uint64), ("n_real_nodes", C.c_uint64), ("node_X", f32p),
                ("node_point", C.POINTER(C.c_uint64)), ("n_polylines", C.c_uint64), ("pl_start", u32p),
                ("pl_end", u32p), ("conn_off", C.POINTER(C.c_uint64)), ("conn_pl", u32p),
                ("n_scene_polylines", C.c_uint64), ("iv_off", C.POINTER(C.c_uint64)), ("iv_start_seg", u32p),
                ("iv_start_xy", f32p), ("iv_end_seg", u32p), ("iv_end_xy", f32p)]


def graph3d_to_dict(g):
    nn, npl, nsp = int(g.n_nodes), int(g.n_polylines), int(g.n_scene_polylines)
    conn_off = as_np(g.conn_off, nn + 1, np.uint64)
    iv_off = as_np(g.iv_off, nsp + 1, np.uint64)
    ni = int(iv_off[-1]) if nsp else 0
    nc = int(conn_off[-1]) if nn else 0
    return {
        "n_nodes": nn, "n_real_nodes": int(g.n_real_nodes), "n_polylines": npl,
        "node_X": as_np(g.node_X, 3 * nn, np.float32).reshape(nn, 3),
        "node_point": as_np(g.node_point, nn, np.uint64),
        "pl_start": as_np(g.pl_start, npl, np.uint32), "pl_end": as_np(g.pl_end, npl, np.uint32),
        "conn_off": conn_off, "conn_pl": as_np(g.conn_pl, nc, np.uint32),
        "iv_off": iv_off, "iv_start_seg": as_np(g.iv_start_seg, ni, np.uint32),
        "iv_start_xy": as_np(g.iv_start_xy, 2 * ni, np.float32).reshape(ni, 2),
        "iv_end_seg": as_np(g.iv_end_seg, ni, np.uint32),
        "iv_end_xy": as_np(g.iv_end_xy, 2 * ni, np.float32).reshape(ni, 2),
    }


class EdgePointsArrays:
    """Owns numpy copies of an edge-point cloud (a dict as returned by edgepoints_to_dict) and
    exposes an EdgePoints struct over them, for the host steps that consume a cloud."""

    def __init__(self, d):
        self.a = {"X": np.ascontiguousarray(d["X"], np.float32), "obs_off": np.ascontiguousarray(d["obs_off"], np.uint64),
                  "obs_view": np.ascontiguousarray(d["obs_view"], np.int32),
                  "obs_pl": np.ascontiguousarray(d["obs_pl"], np.uint32),
                  "obs_seg": np.ascontiguousarray(d["obs_seg"], np.uint32),
                  "obs_xy": np.ascontiguousarray(d["obs_xy"], np.float32),
                  "key": np.ascontiguousarray(d["key"], np.uint32)}
        a = self.a
        n = len(a["obs_off"]) - 1
        self.c = EdgePoints(n, int(a["obs_off"][-1]) if n >= 0 else 0, np_ptr(a["X"], C.c_float),
                            np_ptr(a["obs_off"], C.c_uint64), np_ptr(a["obs_view"], C.c_int32),
                            np_ptr(a["obs_pl"], C.c_uint32), np_ptr(a["obs_seg"], C.c_uint32),
                            np_ptr(a["obs_xy"], C.c_float), np_ptr(a["key"], C.c_uint32), 0, 0, 0, 0, None)


def edgepoints_to_dict(e):
    n, m = int(e.n_points), int(e.n_obs)
    return {
        "n_points": n, "n_obs": m,
        "X": as_np(e.X, 3 * n, np.float32).reshape(n, 3),
        "obs_off": as_np(e.obs_off, n + 1, np.uint64),
        "obs_view": as_np(e.obs_view, m, np.int32),
        "obs_pl": as_np(e.obs_pl, m, np.uint32),
        "obs_seg": as_np(e.obs_seg, m, np.uint32),
        "obs_xy": as_np(e.obs_xy, 2 * m, np.float32).reshape(m, 2),
        "key": as_np(e.key, 4 * n, np.uint32).reshape(n, 4),
        "n_tasks": int(e.n_tasks), "n_hypotheses": int(e.n_hypotheses), "n_chains": int(e.n_chains),
        "flags": int(e.flags),
    }


def candidates_to_dict(c):
    nsv, nt = int(c.n_sv), int(c.n_tasks)
    cand_off = as_np(c.cand_off, nsv + 1, np.uint32)
    start_off = as_np(c.start_off, nsv + 1, np.uint32)
    task_list_off = as_np(c.task_list_off, nt + 1, np.uint32)
    nl = int(task_list_off[-1]) if nt else 0
    list_off = as_np(c.list_off, nl + 1, np.uint32)
    nh = int(list_off[-1]) if nl else 0
    ns = int(start_off[-1]) if nsv else 0
    return {
        "n_sv": nsv, "n_tasks": nt, "cand_off": cand_off,
        "cand_pl": as_np(c.cand_pl, int(cand_off[-1]) if nsv else 0, np.uint32),
        "start_off": start_off, "start_pl": as_np(c.start_pl, ns, np.uint32),
        "start_seg": as_np(c.start_seg, ns, np.uint32),
        "start_xy": as_np(c.start_xy, 2 * ns, np.float32).reshape(ns, 2),
        "task_sv": as_np(c.task_sv, nt, np.uint32), "task_hit": as_np(c.task_hit, nt, np.uint32),
        "task_list_off": task_list_off, "list_off": list_off,
        "hit_pl": as_np(c.hit_pl, nh, np.uint32), "hit_seg": as_np(c.hit_seg, nh, np.uint32),
        "hit_xy": as_np(c.hit_xy, 2 * nh, np.float32).reshape(nh, 2),
    }
