"""-m gpu: the arithmetic contract on the device, bit-for-bit. Runs single
primitives of the product's device headers through the TEST-ONLY probe library
(tests/probe/libeg3d_probe.so, tests/probe/eg3d_probe.h) and compares with IEEE results from numpy
(x86) and with the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

from edgegraph3d_amd import _cdefs as D
from edgegraph3d_amd import api, host

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    assert api.device_count() >= 1, "no HIP device: the product path has no CPU fallback"
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libeg3d_probe.so")
    assert os.path.exists(path), "tests/probe/libeg3d_probe.so is not built (python -m edgegraph3d_amd.build)"
    L = C.CDLL(path)
    L.eg3d_probe_arith.argtypes = [C.c_uint64, D.f64p, D.f64p, D.f64p, D.f64p, D.f32p, D.f32p, D.f32p, D.f32p]
    L.eg3d_probe_triangulate.argtypes = [D.f32p, C.c_int, C.c_uint64, C.c_int, D.i32p, D.f32p, D.f32p, D.u8p, D.f64p]
    s = host.Synth(1)
    yield L, s


def test_ieee_primitives_match_x86_bit_for_bit(ctx):
    c, _ = ctx
    rng = np.random.default_rng(1)
    n = 200000
    a = rng.standard_normal(n) * 10 ** rng.uniform(-8, 8, n)
    b = rng.standard_normal(n) * 10 ** rng.uniform(-8, 8, n)
    cc = rng.standard_normal(n) * 10 ** rng.uniform(-8, 8, n)
    fa, fb, fc = a.astype(np.float32), b.astype(np.float32), cc.astype(np.float32)
    od, of = np.zeros((5, n)), np.zeros((5, n), np.float32)
    rc = c.eg3d_probe_arith(n, D.np_ptr(a, C.c_double), D.np_ptr(b, C.c_double), D.np_ptr(cc, C.c_double),
                                    D.np_ptr(od, C.c_double), D.np_ptr(fa, C.c_float), D.np_ptr(fb, C.c_float),
                                    D.np_ptr(fc, C.c_float), D.np_ptr(of, C.c_float))
    assert rc == 0
    ref_d = [a / b, np.sqrt(np.abs(a)), (a * b) + cc, a.astype(np.float32).astype(np.float64), 1. / np.sqrt(np.abs(b))]
    for i, name in enumerate(["f64 div", "f64 sqrt", "f64 mul then add (no FMA)", "f64->f32 rounding", "f64 1/sqrt"]):
        assert np.array_equal(od[i].view(np.uint64), ref_d[i].view(np.uint64)), name
    dx, dy = (fa - fc).astype(np.float64), (fb - fa).astype(np.float64)
    ref_f = [fa / fb, np.sqrt(np.abs(fa)), (fa * fb) + fc, (dx * dx + dy * dy).astype(np.float32), np.sqrt(np.abs(fa))]
    for i, name in enumerate(["f32 div", "f32 sqrt (EG3D_SQRTF)", "f32 mul then add (no FMA)", "dist2", "__builtin_sqrtf"]):
        assert np.array_equal(of[i].view(np.uint32), ref_f[i].view(np.uint32)), name


def test_triangulation_matches_oracle_including_degenerate_dlt(ctx):
    """(The probe library has ONE DLT form, its compile-time EG3D_DLT_ROWS: the oracle is put in that mode here
    whatever form the suite's product library of this run has.)"""
    import forms
    from oracle import binding as ob
    c, s = ctx
    c.eg3d_probe_dlt_rows.restype = C.c_int
    with forms.oracle_rows(c.eg3d_probe_dlt_rows()):
        _triangulation_check(c, s, ob)


def _triangulation_check(c, s, ob):
    off, view, xy = s.seeds_np()
    cv, cxy = [], []
    for i in range(s.n_seeds):
        a0, k = off[i], off[i + 1] - off[i]
        if k >= 3:
            cv.append(view[a0:a0 + 3]); cxy.append(xy[a0:a0 + 3])
            cv.append(view[a0:a0 + 3][::-1].copy()); cxy.append(xy[a0:a0 + 3][::-1].copy())   # Q11: min id last
    cv = np.ascontiguousarray(np.array(cv, np.int32)); cxy = np.ascontiguousarray(np.array(cxy, np.float32))
    m = len(cv)
    X, val, dlt = np.zeros((m, 3), np.float32), np.zeros(m, np.uint8), np.zeros((m, 3))
    P = np.ascontiguousarray(s.scene_np()["cam_P"])
    rc = c.eg3d_probe_triangulate(D.np_ptr(P, C.c_float), len(P), m, 3, D.np_ptr(cv, C.c_int32), D.np_ptr(cxy, C.c_float),
                                  D.np_ptr(X, C.c_float), D.np_ptr(val, C.c_uint8), D.np_ptr(dlt, C.c_double))
    assert rc == 0
    OL = ob.lib()
    n_deg = 0
    for i in range(m):
        Xo, deg = np.zeros(3, np.float32), C.c_int(0)
        ids = (C.c_int * 3)(*[int(v) for v in cv[i]])
        ok = OL.orc_triangulate(D.np_ptr(P, C.c_float), ids, D.np_ptr(cxy[i], C.c_float), 3, D.np_ptr(Xo, C.c_float), C.byref(deg))
        n_deg += deg.value
        assert bool(ok) == bool(val[i]), i
        if ok:
            assert np.array_equal(Xo.view(np.uint32), X[i].view(np.uint32)), i
        d0 = np.zeros(3)
        mi = int(np.argmin(cv[i]))
        OL.orc_dlt(D.np_ptr(P[cv[i][mi]], C.c_float), D.np_ptr(cxy[i][mi], C.c_float), D.np_ptr(P[cv[i][2]], C.c_float),
                   D.np_ptr(cxy[i][2], C.c_float), D.np_ptr(d0, C.c_double))
        both_nan = np.isnan(d0).all() and np.isnan(dlt[i]).all()
        assert both_nan or np.array_equal(d0.view(np.uint64), dlt[i].view(np.uint64)), i
    assert n_deg > 0
