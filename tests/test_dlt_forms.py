"""The reference pins no OpenCV release, and cv::triangulatePoints built a different linear system
before and after its rewrite: three rows per view (6x4: x*P2-P0, y*P2-P1, x*P1-y*P0; OpenCV
2.4-3.1, the release the reference's README names) or two (4x4, later releases). Both forms are
restated (oracle: orc_set_dlt_rows; product: libeg3d.so = 3 rows, the default since round 3 because the
reference names OpenCV 3.1; libeg3d_dlt4x4.so = 2 rows) and both are kept bit-exact: every `-m gpu` test runs
once per library (tests/conftest.py, tests/forms.py); tools/dlt_form_report.py quantifies how much the choice
changes the output (profiles/r02_dlt_form_report.json)."""
import ctypes as C
import os

import numpy as np
import pytest

from edgegraph3d_amd import host
from oracle import binding as ob
from parity_util import compare_edgepoints

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture
def dlt_rows():
    L = ob.lib()
    before = L.orc_get_dlt_rows()
    yield L
    L.orc_set_dlt_rows(before)


def _dlt(L, P1, xy1, P2, xy2):
    X0 = np.zeros(3, np.float64)
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))
    L.orc_dlt(f(P1), f(xy1), f(P2), f(xy2), X0.ctypes.data_as(C.POINTER(C.c_double)))
    return X0


def _system(P1, xy1, P2, xy2, rows):
    A = []
    for P, (x, y) in ((P1, xy1), (P2, xy2)):
        P = np.asarray(P, np.float64).reshape(4, 4)
        A.append(x * P[2] - P[0])
        A.append(y * P[2] - P[1])
        if rows == 3:
            A.append(x * P[1] - y * P[0])
    return np.array(A)


def test_both_dlt_forms_solve_their_own_system(dlt_rows):
    """Known answer: the homogeneous solution is the right singular vector of the smallest singular
    value of the 4x4 / 6x4 system (numpy SVD as the independent check), rounded to float, then X/w."""
    L = dlt_rows
    rng = np.random.default_rng(5)
    s = host.Synth(0)
    P = s.scene_np()["cam_P"]
    for trial in range(40):
        a, b = rng.choice(len(P), 2, replace=False)
        Xw = rng.uniform(-150, 150, 3)
        obs = []
        for v in (a, b):
            M = P[v].reshape(4, 4).astype(np.float64)
            h = M[:3, :3] @ Xw + M[:3, 3]
            obs.append(np.float32(h[:2] / h[2] + rng.normal(0, 0.7, 2)))
        for rows in (2, 3):
            assert L.orc_set_dlt_rows(rows) == 0
            got = _dlt(L, P[a], obs[0], P[b], obs[1])
            A = _system(P[a], np.float64(obs[0]), P[b], np.float64(obs[1]), rows)
            v = np.linalg.svd(A)[2][-1]
            h = v.astype(np.float32)
            want = (h[:3] / h[3]).astype(np.float64)
            assert np.allclose(got, want, rtol=2e-4, atol=1e-3), (rows, got, want)
    # with noisy observations the two systems have different minimisers: the forms really differ
    L.orc_set_dlt_rows(2)
    x2 = _dlt(L, P[0], obs[0], P[1], obs[1])
    L.orc_set_dlt_rows(3)
    x3 = _dlt(L, P[0], obs[0], P[1], obs[1])
    assert not np.array_equal(x2, x3)


def test_oracle_6x4_form_reproduces_its_fixture(dlt_rows):
    L = dlt_rows
    z = np.load(os.path.join(HERE, "golden", "synthetic_tiny_v1_dlt6x4.npz"))
    want = {k: z["out_" + k] for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")}
    want["n_points"], want["n_obs"] = len(want["X"]), len(want["obs_view"])
    s = host.Synth(0)
    assert L.orc_set_dlt_rows(3) == 0
    r = ob.Oracle(s.scene).match(s.seeds, 0, s.n_seeds, 1)
    rep = compare_edgepoints(want, r)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep
    # and it is NOT the 4x4 result: the Gauss-Newton start decides accept / reject downstream
    assert L.orc_set_dlt_rows(2) == 0
    r2 = ob.Oracle(s.scene).match(s.seeds, 0, s.n_seeds, 1)
    assert r2["n_points"] != r["n_points"] or not np.array_equal(r2["X"], r["X"])


def _project(P, X):
    M = np.asarray(P, np.float64).reshape(4, 4)
    h = M[:3, :3] @ X + M[:3, 3]
    return h[:2] / h[2]


def test_degenerate_pair_start_lies_in_the_numerical_null_space(dlt_rows):
    """Q11: whenever the smallest view id sits LAST in an observation list, em_estimate3Dpositions hands
    cv::triangulatePoints the same camera and the same observation twice (triangulation.cpp:200-217). The system then
    has rank 2 and OpenCV returns SOME vector of the two-dimensional null space — which one is its Jacobi SVD's
    business and cannot be pinned without OpenCV. What CAN be checked independently (numpy's SVD as the judge): the
    start both forms return lies in the numerical null space of its own system, i.e. on the viewing ray of that pixel,
    and the system really has two vanishing singular values."""
    L = dlt_rows
    rng = np.random.default_rng(11)
    s = host.Synth(0)
    P = s.scene_np()["cam_P"]
    not_finite = {2: 0, 3: 0}
    n_trials = 200
    for trial in range(n_trials):
        a = int(rng.integers(len(P)))
        xy = np.float32(rng.uniform([100, 100], [1500, 1100]))
        for rows in (2, 3):
            assert L.orc_set_dlt_rows(rows) == 0
            X0 = _dlt(L, P[a], xy, P[a], xy)
            A = _system(P[a], np.float64(xy), P[a], np.float64(xy), rows)
            sv = np.linalg.svd(A, compute_uv=False)
            assert sv[2] < 1e-9 * sv[0] and sv[3] < 1e-9 * sv[0], (rows, sv)      # rank 2
            if not np.all(np.isfinite(X0)):
                # the 4x4 form's system has EXACTLY repeated rows: two rows of the rotated matrix shrink quadratically
                # to ~1e-160, where this restatement's sqrt(p^2 + beta^2) underflows to 0 (OpenCV calls hypot(), which
                # does not) and the rotation becomes 0/0. The start is then NaN, Gauss-Newton rejects it and the step
                # takes the 3-subset fallback — a documented deviation of the 4x4 form; the 6x4 form of
                # the default library never gets there (its third rows keep the null rows' products away from zero).
                not_finite[rows] += 1
                continue
            h = np.append(X0, 1.0)
            # the homogeneous vector was rounded to float before the division: ~1e-7 relative per component
            assert np.linalg.norm(A @ h) <= 2e-5 * sv[0] * np.linalg.norm(h), (rows, np.linalg.norm(A @ h), sv[0])
            assert np.allclose(_project(P[a], X0), np.float64(xy), atol=2e-2), (rows, _project(P[a], X0), xy)
    assert not_finite[3] == 0, not_finite                 # the default form: always a finite point of the ray
    assert not_finite[2] < 0.7 * n_trials, not_finite     # the 4x4 form: about half (measured 49 %) come out NaN


def test_gauss_newton_reaches_the_least_squares_point_from_either_start(dlt_rows):
    """Cross-check of DLT + Gauss-Newton as a whole against independent numerics: the start from numpy's SVD of the same
    system, the minimiser from scipy's Levenberg-Marquardt on the same reprojection residuals. Wherever the oracle
    accepts a point it must agree with that minimiser to the BASELINE tolerance (1e-4 relative), in both DLT forms —
    so the form (and, more generally, OpenCV's exact SVD output) only matters where a threshold test sits close by."""
    from scipy.optimize import least_squares
    L = dlt_rows
    L.orc_triangulate.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int,
                                  C.POINTER(C.c_float), C.POINTER(C.c_int)]
    rng = np.random.default_rng(23)
    s = host.Synth(0)
    P = s.scene_np()["cam_P"]
    checked = 0
    for trial in range(120):
        k = int(rng.integers(3, min(8, len(P)) + 1))
        views = rng.choice(len(P), k, replace=False).astype(np.int32)
        if trial % 4 == 0:
            views = np.sort(views)[::-1].copy()          # smallest view id LAST: the degenerate start of Q11
        Xw = rng.uniform(-120, 120, 3)
        xy = np.array([_project(P[v], Xw) + rng.normal(0, 0.5, 2) for v in views], np.float32)

        def resid(X):
            return np.concatenate([np.float64(xy[i]) - _project(P[v], X) for i, v in enumerate(views)])

        for rows in (2, 3):
            assert L.orc_set_dlt_rows(rows) == 0
            Pk = np.ascontiguousarray(P.reshape(-1, 16), np.float32)   # the camera table, indexed by view id
            X = np.zeros(3, np.float32)
            deg = C.c_int(0)
            ok = L.orc_triangulate(Pk.ctypes.data_as(C.POINTER(C.c_float)), views.ctypes.data_as(C.POINTER(C.c_int)),
                                   xy.ctypes.data_as(C.POINTER(C.c_float)), k, X.ctypes.data_as(C.POINTER(C.c_float)),
                                   C.byref(deg))
            if not ok:
                continue
            mi, la = int(np.argmin(views)), k - 1
            assert bool(deg.value) == (mi == la)
            # independent start: numpy's null vector of the same system (for the degenerate pair: any point of the ray)
            A = _system(P[views[mi]], np.float64(xy[mi]), P[views[la]], np.float64(xy[la]), rows)
            v = np.linalg.svd(A)[2][-1]
            start = v[:3] / v[3] if abs(v[3]) > 1e-12 else Xw
            best = least_squares(resid, start, method="lm", xtol=1e-14, ftol=1e-14).x
            if np.linalg.norm(resid(best)) ** 2 / (2 * k) > 9 or not np.all(np.isfinite(best)):
                best = least_squares(resid, np.float64(X), method="lm", xtol=1e-14, ftol=1e-14).x
            rel = np.linalg.norm(np.float64(X) - best) / max(np.linalg.norm(best), 1e-9)
            assert rel <= 1e-4, (trial, rows, k, bool(deg.value), X, best, rel)
            checked += 1
    assert checked >= 150
