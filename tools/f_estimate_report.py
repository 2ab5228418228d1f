#!/usr/bin/env python3
"""GPU box: what replacing the analytic fundamental matrices by the build's LMedS estimate from the tracks
 does to the cloud on C2 and C3' -> gpurun_out/f_estimate_report.json."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegraph3d_amd import api, host  # noqa: E402


def line_dist(F, a, b):
    h = np.concatenate([a, np.ones((len(a), 1))], 1)
    l = h @ F.reshape(3, 3).T
    return np.abs(l[:, 0] * b[:, 0] + l[:, 1] * b[:, 1] + l[:, 2]) / np.hypot(l[:, 0], l[:, 1])


out = {}
for name, cfg in (("C2", 2), ("C3'", 3)):
    s = host.Synth(cfg)
    sc = s.scene_np()
    V = sc["n_views"]
    off, view, xy = s.seeds_np()
    F, valid, ncom, failed = host.estimate_F(V, off, view, xy, True, 0xE63D2018)
    ctx = api.Context(s.scene)
    base = ctx.match_refpoints(s.seeds)
    ctx.close()
    sc2 = dict(sc)
    sc2["F"], sc2["F_valid"] = F, valid
    sa = host.SceneArrays(sc2)
    ctx = api.Context(C.byref(sa.c))
    est = ctx.match_refpoints(s.seeds)
    ctx.close()
    # geometric quality of the estimate: epipolar distance of the noise-free projections of the seeds
    Xt = s.seed_truth()
    P = sc["cam_P"].reshape(V, 4, 4).astype(np.float64)
    Xh = np.concatenate([Xt, np.ones((len(Xt), 1))], 1)
    proj = [(Xh @ P[v].T) for v in range(V)]
    proj = [q[:, :2] / q[:, 2:3] for q in proj]
    med = [float(np.median(line_dist(F[i, j], proj[i], proj[j]))) for i in range(V) for j in range(V) if i != j and valid[i, j]]
    out[name] = {"views": V, "pairs_with_matrix": int(valid.sum()), "pairs_total": V * (V - 1), "failed": int(failed),
                 "min_common_points": int(ncom[~np.eye(V, dtype=bool)].min()),
                 "median_epipolar_distance_px_mean_over_pairs": float(np.mean(med)),
                 "median_epipolar_distance_px_worst_pair": float(np.max(med)),
                 "points_analytic_F": int(base["n_points"]), "points_estimated_F": int(est["n_points"]),
                 "observations_analytic_F": int(base["n_obs"]), "observations_estimated_F": int(est["n_obs"])}
    print(name, out[name], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/f_estimate_report.json", "w"), indent=1)
