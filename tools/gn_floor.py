#!/usr/bin/env python3
"""The arithmetic floor of the expand stage's Gauss-Newton work (GPU box).
  1. the product's lane-group solver at FULL density (tests/probe k_probe_gn_dense: windows of seven identical 9-row ADD
     requests = 63 of 64 rows, no request waits for another, 4 waves per SIMD on every CU) -> row-iterations per second: the
     solver's speed of light on this chip under the arithmetic contract (FP64, no FMA, ordered sums);
  2. with a timing build (EG3D_LIB=edgegraph3d_amd/variants/libeg3d_timing.so, `tools/build_variant.sh timing
     -DEG3D_SECTION_TIMING`): the row-iterations one step of the workload really executes (k3b_expand's own counters);
  3. floor = (2) / (1), beside the kernel's measured time.
usage: tools/gn_floor.py [cfg=3] [n_seeds=all] [--json out]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgegraph3d_amd import _cdefs as D, api, build, host

cfg = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0
s = host.Synth(cfg)
sc = s.scene_np()
P = np.ascontiguousarray(sc["cam_P"], np.float32).reshape(-1, 16)
V = P.shape[0]
rng = np.random.default_rng(5)
Xt = s.seed_truth()[0]
views = np.arange(min(9, V), dtype=np.int32)


def project(Pv, X):
    h = Pv.reshape(4, 4).astype(np.float64)[:3] @ np.append(X, 1.0)
    return h[:2] / h[2]


xy = np.array([project(P[v], Xt) + rng.normal(0, 0.3, 2) for v in views], np.float32)


def gn(obs_v, obs_xy, X0):
    """the solver's iteration in plain numpy (not bit-exact, same criterion): (solution, residual passes run)"""
    X, last, passes = np.array(X0, np.float64), 0.0, 0
    n = len(obs_v)
    for _ in range(30):
        passes += 1
        r, J = [], []
        for v, o in zip(obs_v, obs_xy):
            M = P[v].reshape(4, 4).astype(np.float64)[:3]
            h = M @ np.append(X, 1.0)
            r += [o[0] - h[0] / h[2], o[1] - h[1] / h[2]]
            J += [(M[0, :3] * h[2] - M[2, :3] * h[0]) / h[2] ** 2, (M[1, :3] * h[2] - M[2, :3] * h[1]) / h[2] ** 2]
        r, J = np.array(r), np.array(J)
        mse = float(r @ r) / (2 * n)
        if abs(mse - last) < 5e-7:
            break
        last = mse
        X = X + np.linalg.solve(J.T @ J, J.T @ r)
    return X, passes


X8, _ = gn(views[:-1], xy[:-1], Xt + 0.01)          # the point as it stands before the ADD
X0 = X8.astype(np.float32)
_, passes = gn(views, xy, X0.astype(np.float64))     # the ADD solve the probe repeats
build.build_probe()
Lp = C.CDLL(build.PROBE_LIB)
Lp.eg3d_probe_gn_dense.argtypes = [D.f32p, C.c_int, D.i32p, D.f32p, C.c_int, D.f32p, C.c_int, C.c_int, C.POINTER(C.c_float), D.f32p]
n = len(views)
out = {"workload_cfg": cfg, "request": {"rows": int(n), "residual_passes_per_solve": passes}, "dense": []}
best = 0.0
for waves_per_cu, rounds in ((16, 300), (32, 200), (64, 100)):
    blocks = 256 * waves_per_cu
    ms, xo = C.c_float(0), np.zeros(2, np.float32)
    rc = Lp.eg3d_probe_gn_dense(D.np_ptr(P.reshape(-1), C.c_float), V, D.np_ptr(views, C.c_int32), D.np_ptr(xy.reshape(-1), C.c_float), n,
                                D.np_ptr(X0, C.c_float), blocks, rounds, C.byref(ms), D.np_ptr(xo, C.c_float))
    assert rc == 0
    rate = blocks * rounds * 7 * n * passes / (ms.value * 1e-3)
    best = max(best, rate)
    out["dense"].append({"blocks": blocks, "rounds_per_block": rounds, "ms": ms.value, "row_iterations_per_s": rate,
                         "round_iterations_per_s": blocks * rounds * passes / (ms.value * 1e-3), "accepted_solves_of_block0_lane0": float(xo[1])})
    print("dense solver: %5d blocks x %3d windows of 7 x %d rows, %d passes each: %.2f ms -> %.3e row-iterations/s" % (blocks, rounds, n, passes, ms.value, rate), flush=True)
out["dense_peak_row_iterations_per_s"] = best
L = api.lib()
if hasattr(L, "eg3d_probe_gn"):
    ctx = api.Context(s.scene); ctx.upload_seeds(s.seeds)
    buf = (C.c_ulonglong * 128)()
    ns = nseeds or s.n_seeds
    ctx.match_resident(0, ns, device_only=True)
    L.eg3d_probe_gn(buf, 1)
    r = ctx.match_resident(0, ns, device_only=True)
    L.eg3d_probe_gn(buf, 1)
    b = list(buf)
    out["step"] = {"solves": b[64], "rows": b[66], "row_iterations": b[68], "lane_iterations_held": b[67], "rounds": b[69],
                   "row_fill_of_64": 64.0 * b[68] / max(1, b[67]), "edge_points": int(r["n_points"]), "seeds": ns, "long_solves": b[102], "long_rounds": b[103]}
    out["floor_ms_at_dense_peak"] = 1e3 * b[68] / best
    print("one step of cfg %d: %d solves, %d row-iterations (fill %.1f of 64) -> %.2f ms at the dense peak"
          % (cfg, b[64], b[68], out["step"]["row_fill_of_64"], out["floor_ms_at_dense_peak"]))
else:
    print("(no timing build loaded: set EG3D_LIB to a -DEG3D_SECTION_TIMING variant for the step's row-iteration count)")
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
