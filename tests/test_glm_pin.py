"""Pins of the hot path's glm-based arithmetic against the REFERENCE'S OWN vendored glm (external/glm 0.9.6, header-only —
the one part of the reference's arithmetic that builds in this image; OpenCV's part cannot be pinned).

tests/golden/glm_pin_v1.npz holds inputs and the outputs glm itself computed (tests/golden/make_glm_golden.py, driver
tests/glm/glm_driver.cpp). Checked bit for bit against them:
  * the oracle's hand-written evaluation orders: compute_projection (geometric_utilities.cpp:973-977),
    compute_anglecos (:579-618), minimum_distancesq (:940-954 with squared_2d_distance :555-557);
  * the product's host camera model (OpenMvgParser.cpp:289 t = -center * rotation, :107-125 cameraMatrix = eMatrix * kMatrix);
  * on the GPU: project_f32, seg_line_cos, seg_closest of the product's device headers.
Where the reference tree is present (the build container) the same comparison also runs LIVE on ~1 M fresh random cases
per function. Values are compared as bit patterns; NaN results only have to be NaN on both sides.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_glm_golden as G  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "glm_pin_v1.npz")
f32p = C.POINTER(C.c_float)


def _ptr(a):
    return a.ctypes.data_as(f32p)


def _same(got, want, what):
    got = np.ascontiguousarray(got, np.float32).reshape(-1)
    want = np.ascontiguousarray(want, np.float32).reshape(-1)
    both_nan = np.isnan(got) & np.isnan(want)
    eq = (got.view(np.uint32) == want.view(np.uint32)) | both_nan
    bad = np.flatnonzero(~eq)
    assert bad.size == 0, "%s: %d of %d values differ from glm, first at %d: %r vs %r" % (
        what, bad.size, got.size, bad[0], got[bad[0]], want[bad[0]])


def _oracle():
    from oracle import binding as ob
    L = ob.lib()
    L.orc_batch_project.argtypes = [C.c_uint64, f32p, f32p, f32p]
    L.orc_batch_project.restype = None
    L.orc_batch_mindist.argtypes = [C.c_uint64, f32p, f32p]
    L.orc_batch_mindist.restype = None
    L.orc_batch_anglecos.argtypes = [C.c_uint64, f32p, f32p]
    L.orc_batch_anglecos.restype = None
    return L


def _oracle_eval(mode, a):
    L = _oracle()
    a = np.ascontiguousarray(a, np.float32)
    n = a.shape[0]
    if mode == "proj":
        P = np.ascontiguousarray(a[:, :16])
        X = np.ascontiguousarray(a[:, 16:19])
        out = np.zeros((n, 2), np.float32)
        L.orc_batch_project(n, _ptr(P), _ptr(X), _ptr(out))
        return out
    if mode == "mindist":
        out = np.zeros((n, 3), np.float32)
        L.orc_batch_mindist(n, _ptr(a), _ptr(out))
        return out
    out = np.zeros((n, 1), np.float32)
    L.orc_batch_anglecos(n, _ptr(a), _ptr(out))
    return out


def _product_cam(a):
    from edgegraph3d_amd import host
    H = host.lib()
    H.eg3d_host_camera_model.argtypes = [C.c_uint64, f32p, f32p, f32p, f32p, f32p]
    a = np.ascontiguousarray(a, np.float32)
    n = a.shape[0]
    fpp, R, Cc = (np.ascontiguousarray(a[:, :3]), np.ascontiguousarray(a[:, 3:12]), np.ascontiguousarray(a[:, 12:15]))
    t = np.zeros((n, 3), np.float32)
    P = np.zeros((n, 16), np.float32)
    assert H.eg3d_host_camera_model(n, _ptr(fpp), _ptr(R), _ptr(Cc), _ptr(t), _ptr(P)) == 0
    return np.concatenate([t, P], 1)


def _glm_want(mode, out):
    return out[:, :2] if mode == "proj" else out


def test_oracle_evaluation_orders_equal_glm_on_the_committed_vectors():
    g = np.load(GOLD)
    for mode in ("proj", "anglecos", "mindist"):
        _same(_oracle_eval(mode, g[mode + "_in"]), _glm_want(mode, g[mode + "_out"]), "oracle " + mode)


def test_product_camera_model_equals_glm_on_the_committed_vectors():
    g = np.load(GOLD)
    _same(_product_cam(g["cam_in"]), g["cam_out"], "host camera model (t, P)")
    # the last row of the composed matrix is all zero (Q6) and the fixture says so too
    assert not g["cam_out"][:-4, 15:19].any()


@pytest.mark.skipif(not G.have_reference(), reason="the reference tree is not present (build container only)")
def test_live_one_million_cases_per_function_against_the_reference_glm():
    n = 1 << 20
    for i, mode in enumerate(("proj", "anglecos", "mindist")):
        a = G.cases(mode, n, 0xA11CE + i)
        _same(_oracle_eval(mode, a), _glm_want(mode, G.run_glm(mode, a)), "oracle %s (live, %d cases)" % (mode, n))
    a = G.cases("cam", 1 << 18, 0xCA3)
    _same(_product_cam(a), G.run_glm("cam", a), "host camera model (live)")


@pytest.mark.skipif(not G.have_reference(), reason="the reference tree is not present (build container only)")
def test_committed_vectors_are_what_the_reference_glm_computes_now():
    """The fixture is regenerated from its recorded inputs and must come out identical (it is data, not a copy of code)."""
    g = np.load(GOLD)
    for mode in ("cam", "proj", "anglecos", "mindist"):
        _same(G.run_glm(mode, g[mode + "_in"]), g[mode + "_out"], "fixture " + mode)


@pytest.mark.gpu
def test_device_geometry_primitives_equal_glm_on_the_committed_vectors():
    from edgegraph3d_amd import build as b
    P = C.CDLL(b.build_probe())
    P.eg3d_probe_geom.argtypes = [C.c_uint64, C.c_int, f32p, f32p]
    g = np.load(GOLD)
    for k, (mode, width) in enumerate((("proj", 2), ("anglecos", 1), ("mindist", 3))):
        a = np.ascontiguousarray(g[mode + "_in"], np.float32)
        out = np.zeros((a.shape[0], width), np.float32)
        assert P.eg3d_probe_geom(a.shape[0], k, _ptr(a), _ptr(out)) == 0
        _same(out, _glm_want(mode, g[mode + "_out"]), "device " + mode)
