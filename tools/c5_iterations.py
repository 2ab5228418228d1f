#!/usr/bin/env python3
"""Config 5 (k5_gn_filter): how many residual passes each of the 1 M points runs (the kernel's own per-point routine compiled
for the host: tests/hostsim), and what that means for a kernel that gives one lane to a point: a wavefront runs until its
slowest lane is done, so its active-lane fraction is mean(passes of its 64 points) / max(passes of its 64 points).
CPU only. usage: c5_iterations.py [n_points] [out.json]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from edgegraph3d_amd import _cdefs as D, host  # noqa: E402
import hostsim_binding as hs  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1000000
s = host.Synth(5)
X, off, view, xy = s.points(n)
L = hs.lib()
L.hostsim_gn_filter_iters.argtypes = [D.f32p, D.f32p, D.u32p, D.i32p, D.f32p, C.c_uint64, C.c_float, C.c_int, D.u8p]
camP = np.ascontiguousarray(s.scene_np()["cam_P"], np.float32)
out = {}
for legacy in (0, 1):
    it = np.zeros(n, np.uint8)
    L.hostsim_gn_filter_iters(D.np_ptr(camP, C.c_float), D.np_ptr(np.ascontiguousarray(X, np.float32), C.c_float),
                              D.np_ptr(np.ascontiguousarray(off, np.uint32), C.c_uint32), D.np_ptr(np.ascontiguousarray(view, np.int32), C.c_int32),
                              D.np_ptr(np.ascontiguousarray(xy, np.float32), C.c_float), n, 2.25, legacy, D.np_ptr(it, C.c_uint8))
    hist = np.bincount(it, minlength=31)[1:31]
    k = np.diff(off.astype(np.int64))
    work = it.astype(np.int64) * k                      # residual rows a point evaluates (per pass: its k observations)
    nw = (n // 64) * 64
    w_it = it[:nw].reshape(-1, 64).astype(np.float64)
    w_work = work[:nw].reshape(-1, 64).astype(np.float64)
    # a wave's time ~ sum over passes of the largest k among the lanes still active; lower bound used here: max over lanes of passes x k
    lane_frac_passes = float(w_it.mean(axis=1).sum() / w_it.max(axis=1).sum())
    lane_frac_rows = float(w_work.sum() / (64.0 * w_work.max(axis=1)).sum())
    out["legacy_abs" if legacy else "default"] = {
        "passes_histogram_1_to_30": [int(x) for x in hist], "mean_passes": float(it.mean()), "median_passes": float(np.median(it)),
        "points_running_all_30": int(hist[29]), "active_lane_fraction_if_one_lane_per_point": {
            "by_passes": lane_frac_passes, "by_rows": lane_frac_rows,
            "what": "mean / max over the 64 points of a wavefront, summed over wavefronts (points in input order, as the kernel takes them)"}}
out["n_points"] = n
out["mean_observations"] = float(np.diff(off.astype(np.int64)).mean())
txt = json.dumps(out, indent=1)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
