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
