"""Python view of the C ABI (include/eg3d.h) in libeg3d.so — plumbing over ctypes.

There is no CPU fallback: constructing a Context without the built HIP library or without a
GPU raises. The class mirrors the reference call surface for the path:
  Context(scene)                     ~ PLGEdgeManager / PLGPCM3ViewsPLGFollowing construction
  Context.match_refpoints(seeds)     ~ plg_matching_from_refpoints_parallel(sfmd, em, cm, plgmm)
  Context.candidates(seeds)          ~ PLGEdgeManager::detect_nearby_intersections_and_correspondences_plgp
  Context.gn_filter(...)             ~ gaussNewtonFiltering(sfmd, inliers, gn_max_mse)
"""
import ctypes as C
import os

import numpy as np

from . import _cdefs as D

_LIB = None


class Eg3dError(RuntimeError):
    pass


def lib_path():
    # EG3D_LIB selects an alternative build of the same library (tuning experiments only)
    return os.environ.get("EG3D_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libeg3d.so")


def lib():
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise Eg3dError("libeg3d.so (HIP extension) is missing: run `python -m edgegraph3d_amd.build`. "
                            "There is no CPU fallback.")
        L = C.CDLL(path)
        L.eg3d_last_error.restype = C.c_char_p
        L.eg3d_device_count.restype = C.c_int
        L.eg3d_dlt_rows.restype = C.c_int
        L.eg3d_create.argtypes = [C.POINTER(D.Scene), C.c_int, C.POINTER(C.c_void_p)]
        L.eg3d_destroy.argtypes = [C.c_void_p]
        L.eg3d_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.eg3d_get_grid.argtypes = [C.c_void_p, C.c_int, C.c_int, D.u32p, D.u32p, C.POINTER(D.u32p), C.POINTER(D.u32p)]
        L.eg3d_candidates_run.argtypes = [C.c_void_p, C.POINTER(D.Seeds), C.c_uint32, C.c_uint32, C.POINTER(D.Candidates)]
        L.eg3d_free_candidates.argtypes = [C.POINTER(D.Candidates)]
        L.eg3d_match_refpoints.argtypes = [C.c_void_p, C.POINTER(D.Seeds), C.c_uint32, C.c_uint32, C.c_int,
                                           C.POINTER(D.EdgePoints), C.POINTER(D.StageTimes)]
        L.eg3d_free_edgepoints.argtypes = [C.POINTER(D.EdgePoints)]
        L.eg3d_upload_seeds.argtypes = [C.c_void_p, C.POINTER(D.Seeds)]
        L.eg3d_match_resident.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(D.EdgePoints),
                                          C.POINTER(D.StageTimes)]
        L.eg3d_match_polyline_sets.argtypes = [C.c_void_p, C.POINTER(D.PolylineSets), C.c_uint32, C.c_uint32, C.c_int,
                                               C.POINTER(D.EdgePoints), C.POINTER(D.StageTimes)]
        L.eg3d_last_device_output.argtypes = [C.c_void_p, C.POINTER(D.DeviceEdgePoints)]
        L.eg3d_set_pipelining.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.eg3d_gn_filter.argtypes = [C.c_void_p, D.f32p, D.u32p, D.i32p, D.f32p, C.c_uint64, C.c_float, C.c_int,
                                     D.f32p, D.u8p, D.f32p]
        _LIB = L
    return _LIB


# every symbol include/eg3d.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = [
    "eg3d_last_error", "eg3d_device_count", "eg3d_dlt_rows", "eg3d_create", "eg3d_clone", "eg3d_destroy", "eg3d_get_grid", "eg3d_candidates_run",
    "eg3d_free_candidates", "eg3d_match_refpoints", "eg3d_free_edgepoints", "eg3d_upload_seeds",
    "eg3d_match_resident", "eg3d_gn_filter", "eg3d_last_device_output", "eg3d_match_polyline_sets", "eg3d_set_pipelining",
]


def _check(rc, what):
    if rc != 0:
        raise Eg3dError("%s failed (rc=%d): %s" % (what, rc, lib().eg3d_last_error().decode()))


def device_count():
    return int(lib().eg3d_device_count())


class Context:
    def __init__(self, scene_ptr, device=0, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            _check(lib().eg3d_create(scene_ptr, device, C.byref(self._h)), "eg3d_create")

    def clone(self):
        """A context sharing this one's scene and resident seeds (own stream and work buffers)."""
        h = C.c_void_p()
        _check(lib().eg3d_clone(self._h, C.byref(h)), "eg3d_clone")
        return Context(None, _handle=h)

    def set_pipelining(self, lanes=0, units=0):
        """Sub-batches of one call kept in flight inside the library (eg3d_set_pipelining): lanes=1 switches it off."""
        _check(lib().eg3d_set_pipelining(self._h, lanes, units), "eg3d_set_pipelining")

    def close(self):
        if self._h:
            lib().eg3d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def grid(self, view, which):
        ncols, nrows = C.c_uint32(), C.c_uint32()
        off, ids = D.u32p(), D.u32p()
        _check(lib().eg3d_get_grid(self._h, view, which, C.byref(ncols), C.byref(nrows), C.byref(off), C.byref(ids)),
               "eg3d_get_grid")
        n = ncols.value * nrows.value
        o = D.as_np(off, n + 1, np.uint32)
        return ncols.value, nrows.value, o, D.as_np(ids, int(o[-1]), np.uint32)

    def upload_seeds(self, seeds_ptr):
        _check(lib().eg3d_upload_seeds(self._h, seeds_ptr), "eg3d_upload_seeds")

    def match_resident(self, begin, end, device_only=False):
        e, tm = D.EdgePoints(), D.StageTimes()
        rc = lib().eg3d_match_resident(self._h, begin, end, 1 if device_only else 0, C.byref(e), C.byref(tm))
        _check(rc, "eg3d_match_resident")
        if device_only:
            d = {"n_points": int(e.n_points), "n_obs": int(e.n_obs), "n_tasks": int(e.n_tasks),
                 "n_hypotheses": int(e.n_hypotheses), "n_chains": int(e.n_chains), "flags": int(e.flags)}
        else:
            d = D.edgepoints_to_dict(e)
        lib().eg3d_free_edgepoints(C.byref(e))
        d["times"] = {f[0]: getattr(tm, f[0]) for f in D.StageTimes._fields_}
        return d

    def time_match_to_host(self, begin, end):
        """Wall seconds of ONE eg3d_match_resident(..., device_only=0) call at the C ABI (kernels + D2H of the
        cloud into caller-owned arrays), without this wrapper's conversion to numpy; returns (seconds, n_points)."""
        import time
        e, tm = D.EdgePoints(), D.StageTimes()
        t0 = time.perf_counter()
        rc = lib().eg3d_match_resident(self._h, begin, end, 0, C.byref(e), C.byref(tm))
        dt = time.perf_counter() - t0
        _check(rc, "eg3d_match_resident")
        n = int(e.n_points)
        lib().eg3d_free_edgepoints(C.byref(e))
        return dt, n

    def time_match_sets_to_host(self, n_sets, row_off, pl_ids):
        """The same for eg3d_match_polyline_sets: (seconds, n_points) of one call with device_only=0 at the C ABI."""
        import time
        row_off = np.ascontiguousarray(row_off, np.uint32)
        pl_ids = np.ascontiguousarray(pl_ids if len(pl_ids) else [0], np.uint32)
        ps = D.PolylineSets(n_sets, D.np_ptr(row_off, C.c_uint32), D.np_ptr(pl_ids, C.c_uint32))
        e, tm = D.EdgePoints(), D.StageTimes()
        t0 = time.perf_counter()
        rc = lib().eg3d_match_polyline_sets(self._h, C.byref(ps), 0, n_sets, 0, C.byref(e), C.byref(tm))
        dt = time.perf_counter() - t0
        _check(rc, "eg3d_match_polyline_sets")
        n = int(e.n_points)
        lib().eg3d_free_edgepoints(C.byref(e))
        return dt, n

    def match_refpoints(self, seeds_ptr, begin=0, end=None, device_only=False):
        if end is None:
            end = int(seeds_ptr.contents.n_seeds) if hasattr(seeds_ptr, "contents") else int(seeds_ptr.n_seeds)
        self.upload_seeds(seeds_ptr)
        return self.match_resident(begin, end, device_only)

    def match_polyline_sets(self, n_sets, row_off, pl_ids, begin=0, end=None, device_only=False):
        """Pipelines 1-2 extractor: sets = CSR over rows (set * V + view) of polyline ids."""
        if end is None:
            end = n_sets
        row_off = np.ascontiguousarray(row_off, np.uint32)
        pl_ids = np.ascontiguousarray(pl_ids if len(pl_ids) else [0], np.uint32)
        ps = D.PolylineSets(n_sets, D.np_ptr(row_off, C.c_uint32), D.np_ptr(pl_ids, C.c_uint32))
        e, tm = D.EdgePoints(), D.StageTimes()
        rc = lib().eg3d_match_polyline_sets(self._h, C.byref(ps), begin, end, 1 if device_only else 0, C.byref(e),
                                            C.byref(tm))
        _check(rc, "eg3d_match_polyline_sets")
        if device_only:
            d = {"n_points": int(e.n_points), "n_obs": int(e.n_obs), "n_tasks": int(e.n_tasks),
                 "n_hypotheses": int(e.n_hypotheses), "n_chains": int(e.n_chains), "flags": int(e.flags)}
        else:
            d = D.edgepoints_to_dict(e)
        lib().eg3d_free_edgepoints(C.byref(e))
        d["times"] = {f[0]: getattr(tm, f[0]) for f in D.StageTimes._fields_}
        return d

    def last_device_output(self):
        d = D.DeviceEdgePoints()
        _check(lib().eg3d_last_device_output(self._h, C.byref(d)), "eg3d_last_device_output")
        return d

    def fetch_device_output(self):
        """Test/bench plumbing: copies the cloud eg3d_last_device_output views (HBM) into numpy arrays shaped like
        a host result (obs_off gets its final n_obs sentinel). Needs `complete`."""
        d = self.last_device_output()
        if not d.complete:
            raise RuntimeError("the device view does not hold the whole cloud of the last call")
        try:
            hip = C.CDLL("libamdhip64.so.7")      # by SONAME: the copy libeg3d.so (or torch) already loaded
        except OSError:
            hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def fetch(ptr, n, dtype):
            a = np.empty(n, dtype)
            if n and hip.hipMemcpy(a.ctypes.data, C.cast(ptr, C.c_void_p), a.nbytes, 2) != 0:
                raise RuntimeError("hipMemcpy of the device cloud failed")
            return a
        n, m = int(d.n_points), int(d.n_obs)
        return {"n_points": n, "n_obs": m, "X": fetch(d.X, 3 * n, np.float32).reshape(n, 3),
                "obs_off": np.concatenate([fetch(d.obs_off, n, np.uint64), np.array([m], np.uint64)]),
                "key": fetch(d.key, 4 * n, np.uint32).reshape(n, 4), "obs_view": fetch(d.obs_view, m, np.int32),
                "obs_pl": fetch(d.obs_pl, m, np.uint32), "obs_seg": fetch(d.obs_seg, m, np.uint32),
                "obs_xy": fetch(d.obs_xy, 2 * m, np.float32).reshape(m, 2)}

    def fetch_device_points(self, dev, p0, p1):
        """Test plumbing: points [p0, p1) of a device-resident cloud `dev` (a DeviceEdgePoints: this context's last
        output, or the result of a gather / concat) as numpy arrays, observation offsets rebased to the slice.
        For clouds too large to copy whole (BASELINE config 4 in one call: 82 GB)."""
        try:
            hip = C.CDLL("libamdhip64.so.7")
        except OSError:
            hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

        def fetch(ptr, first, n, dtype):
            a = np.empty(n, dtype)
            src = C.cast(ptr, C.c_void_p).value + first * a.itemsize if n else 0
            if n and hip.hipMemcpy(a.ctypes.data, C.c_void_p(src), a.nbytes, 2) != 0:
                raise RuntimeError("hipMemcpy of the device cloud failed")
            return a
        n_all, m_all = int(dev.n_points), int(dev.n_obs)
        n = p1 - p0
        off = fetch(dev.obs_off, p0, n + (1 if p1 < n_all else 0), np.uint64)
        if p1 >= n_all:
            off = np.concatenate([off, np.array([m_all], np.uint64)])
        o0, o1 = (int(off[0]), int(off[-1])) if n else (0, 0)
        m = o1 - o0
        return {"n_points": n, "n_obs": m, "X": fetch(dev.X, 3 * p0, 3 * n, np.float32).reshape(n, 3),
                "obs_off": off - np.uint64(o0), "key": fetch(dev.key, 4 * p0, 4 * n, np.uint32).reshape(n, 4),
                "obs_view": fetch(dev.obs_view, o0, m, np.int32), "obs_pl": fetch(dev.obs_pl, o0, m, np.uint32),
                "obs_seg": fetch(dev.obs_seg, o0, m, np.uint32),
                "obs_xy": fetch(dev.obs_xy, 2 * o0, 2 * m, np.float32).reshape(m, 2)}

    def candidates(self, seeds_ptr, begin, end):
        c = D.Candidates()
        _check(lib().eg3d_candidates_run(self._h, seeds_ptr, begin, end, C.byref(c)), "eg3d_candidates_run")
        d = D.candidates_to_dict(c)
        lib().eg3d_free_candidates(C.byref(c))
        return d

    def gn_filter(self, X, obs_off, obs_view, obs_xy, gn_max_mse=2.25, legacy_abs=False):
        X = np.ascontiguousarray(X, np.float32)
        obs_off = np.ascontiguousarray(obs_off, np.uint32)
        obs_view = np.ascontiguousarray(obs_view, np.int32)
        obs_xy = np.ascontiguousarray(obs_xy, np.float32)
        n = len(obs_off) - 1
        Xo = np.zeros((n, 3), np.float32)
        inl = np.zeros(n, np.uint8)
        ms = C.c_float(0)
        _check(lib().eg3d_gn_filter(self._h, D.np_ptr(X, C.c_float), D.np_ptr(obs_off, C.c_uint32),
                                    D.np_ptr(obs_view, C.c_int32), D.np_ptr(obs_xy, C.c_float), n, gn_max_mse,
                                    1 if legacy_abs else 0, D.np_ptr(Xo, C.c_float), D.np_ptr(inl, C.c_uint8),
                                    C.byref(ms)), "eg3d_gn_filter")
        return Xo, inl, ms.value
