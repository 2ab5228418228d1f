"""fundamental matrices from the point tracks (geometric_utilities.cpp:754-820).
The deterministic part — which ordered pairs get a matrix (>= 10 common points), the common-point
order and the position rule (last observation of a view wins) — is checked against the oracle's
reference-shaped restatement. The estimate itself replaces cv::findFundamentalMat(FM_LMEDS), a
randomised OpenCV routine that cannot be reproduced here; it is checked geometrically against the
analytic matrices of the generator's cameras (epipolar distance of the noise-free correspondences)."""
import ctypes as C

import numpy as np

from edgegraph3d_amd import _cdefs as D
from edgegraph3d_amd import host
from oracle import binding as ob


def _orc_pair(V, off, view, xy, i, j):
    L = ob.lib()
    L.orc_pair_correspondences.restype = C.c_int
    L.orc_pair_correspondences.argtypes = [C.c_int, C.c_uint64, D.u32p, D.i32p, D.f32p, C.c_int, C.c_int, D.u32p,
                                           D.f32p, D.f32p, D.u32p]
    n = len(off) - 1
    ids = np.zeros(n, np.uint32)
    a = np.zeros((n, 2), np.float32)
    b = np.zeros((n, 2), np.float32)
    nc = C.c_uint32()
    m = L.orc_pair_correspondences(V, n, D.np_ptr(off, C.c_uint32), D.np_ptr(view, C.c_int32),
                                   D.np_ptr(np.ascontiguousarray(xy, np.float32), C.c_float), i, j,
                                   D.np_ptr(ids, C.c_uint32), D.np_ptr(a, C.c_float), D.np_ptr(b, C.c_float), C.byref(nc))
    return m, nc.value, ids[:m], a[:m], b[:m]


def _line_dist(F, a, b):
    """distance of b (view j) from the epipolar line F a of a (view i), per row"""
    h = np.concatenate([a, np.ones((len(a), 1))], 1)
    l = h @ F.reshape(3, 3).T
    return np.abs((l[:, 0] * b[:, 0] + l[:, 1] * b[:, 1] + l[:, 2])) / np.hypot(l[:, 0], l[:, 1])


def test_validity_rule_and_counts_match_the_oracle():
    s = host.Synth(0)  # 4 views, short tracks: a mix of pairs above and below 10 common points
    off, view, xy = s.seeds_np()
    V = s.scene_np()["n_views"]
    # thin the tracks so that some pairs fall below the limit, and repeat a view id inside one track (Q2)
    keep = np.ones(len(view), bool)
    rng = np.random.default_rng(5)
    for p in range(len(off) - 1):
        if p % 3:
            k = rng.integers(off[p], off[p + 1])
            keep[k] = False
    new_off = np.zeros_like(off)
    new_off[1:] = np.cumsum([keep[off[p]:off[p + 1]].sum() for p in range(len(off) - 1)])
    view2, xy2 = view[keep].copy(), xy[keep].copy()
    if new_off[1] - new_off[0] >= 2:
        view2[new_off[0] + 1] = view2[new_off[0]]  # repeated view id: the later observation is the one used
    _, valid, ncom, _ = host.estimate_F(V, new_off, view2, xy2, estimate=False)
    for i in range(V):
        for j in range(V):
            if i == j:
                assert valid[i, j] == 0 and ncom[i, j] == 0
                continue
            m, nc, ids, a, b = _orc_pair(V, new_off, view2, xy2, i, j)
            assert ncom[i, j] == nc
            assert valid[i, j] == (1 if nc >= 10 else 0)
            assert m == (nc if nc >= 10 else 0)
    assert valid.any()


def test_estimate_agrees_with_the_cameras_geometrically():
    s = host.Synth(2)  # C2: 8 views, 2000 seeds, observation noise 0.4 px
    sc = s.scene_np()
    V = sc["n_views"]
    off, view, xy = s.seeds_np()
    F, valid, ncom, failed = host.estimate_F(V, off, view, xy, estimate=True, rng_seed=7)
    assert failed == 0
    F2, valid2, _, _ = host.estimate_F(V, off, view, xy, estimate=True, rng_seed=7)
    assert np.array_equal(F, F2) and np.array_equal(valid, valid2)  # deterministic in the seed
    # noise-free correspondences: the true 3-D seed positions projected with the generator's cameras
    Xt = s.seed_truth()
    P = sc["cam_P"].reshape(V, 4, 4).astype(np.float64)
    Xh = np.concatenate([Xt, np.ones((len(Xt), 1))], 1)
    proj = []
    for v in range(V):
        q = Xh @ P[v].T
        proj.append(q[:, :2] / q[:, 2:3])
    worst = 0.0
    checked = 0
    for i in range(V):
        for j in range(V):
            if i == j or not valid[i, j]:
                continue
            assert sc["F_valid"][i, j]
            inside = np.all((proj[i] > 0) & (proj[i] < [sc["width"], sc["height"]]) & (proj[j] > 0)
                            & (proj[j] < [sc["width"], sc["height"]]), axis=1)
            d_est = _line_dist(F[i, j], proj[i][inside], proj[j][inside])
            d_ana = _line_dist(sc["F"][i, j], proj[i][inside], proj[j][inside])
            assert np.median(d_ana) < 1e-2
            worst = max(worst, float(np.median(d_est)))
            checked += 1
    assert checked >= V * (V - 1) // 2
    assert worst < 1.5, worst  # observation noise is 0.4 px; the analytic matrices give ~0
