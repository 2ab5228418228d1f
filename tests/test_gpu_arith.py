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
    L.eg3d_probe_dlt_groups.argtypes = [D.f32p, C.c_int, C.c_uint64, C.c_int, D.i32p, D.f32p, D.f64p]
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
    # the lane-group form of the same decomposition (dlt2_grp8: what k3b_expand runs) against the one-lane form just
    # checked against the oracle: every case, degenerate pairs included, bit for bit
    dlt_g = np.zeros((m, 3))
    rc = c.eg3d_probe_dlt_groups(D.np_ptr(P, C.c_float), len(P), m, 3, D.np_ptr(cv, C.c_int32), D.np_ptr(cxy, C.c_float),
                                 D.np_ptr(dlt_g, C.c_double))
    assert rc == 0
    same = (dlt_g.view(np.uint64) == dlt.view(np.uint64)).all(axis=1) | (np.isnan(dlt_g).all(axis=1) & np.isnan(dlt).all(axis=1))
    assert same.all(), np.flatnonzero(~same)[:10]


def test_shared_reciprocal_divisions_equal_plain_divisions(ctx):
    """eg3d_dev_coopgn.h: a Gauss-Newton row divides two numbers by zH and six by zH^2 through TWO refined reciprocals
    (rcp + 4 fma per divisor, mul + 2 fma per numerator) instead of eight full division sequences — only when a range
    test guarantees that the full sequence would not rescale its operands. (1) On operand pairs inside the admitted
    regime — divisor 2^-200..2^200 (zH^2), numerator zero or 2^-253..2^202 — the short form equals `num / den` bit for
    bit, signs of zero included; (2) whole rows through the guarded path equal the plain rows bit for bit on realistic
    AND hostile inputs (tiny / huge / zero / NaN coordinates, cameras scaled by 1e+-24 or all zero: the guard must send
    whatever leaves the regime to the plain divisions)."""
    c, _ = ctx
    c.eg3d_probe_gn_div.argtypes = [C.c_uint64, D.f64p, D.f64p, D.f64p]
    c.eg3d_probe_gn_rows.argtypes = [C.c_uint64, D.f32p, D.f32p, D.f64p, D.f64p]
    rng = np.random.default_rng(7)
    n = 2000000
    den = rng.standard_normal(n) * 2.0 ** rng.uniform(-200, 200, n)
    den[np.abs(den) < 2.0 ** -200] = 1.5
    num = rng.standard_normal(n) * 2.0 ** rng.uniform(-253, 202, n)
    num[: n // 50] = 0.0
    num[n // 50: n // 25] = -0.0
    num[n // 25: n // 20] = den[n // 25: n // 20]                      # quotient exactly 1
    num[n // 20: n // 10] = den[n // 20: n // 10] * rng.integers(1, 1 << 20, n // 10 - n // 20)   # exact quotients
    # numbers one ulp around representable quotients (rounding ties of the division)
    num[n // 10: n // 5] = np.nextafter(num[n // 10: n // 5], np.inf)
    out = np.zeros(3 * n)
    assert c.eg3d_probe_gn_div(n, D.np_ptr(num, C.c_double), D.np_ptr(den, C.c_double), D.np_ptr(out, C.c_double)) == 0
    plain, fast = out[:n].view(np.uint64), out[n:2 * n].view(np.uint64)
    bad = np.nonzero(plain != fast)[0]
    assert len(bad) == 0, (len(bad), num[bad[:4]], den[bad[:4]], out[:n][bad[:4]], out[n:2 * n][bad[:4]])
    assert np.array_equal(out[:n].view(np.uint64), (num / den).view(np.uint64))   # and both are the IEEE quotient
    # ---- whole rows
    s = host.Synth(4)
    P = s.scene_np()["cam_P"].reshape(-1, 16)
    m = 400000
    Pi = np.ascontiguousarray(P[rng.integers(0, len(P), m)], np.float32)
    X = rng.uniform(-150, 150, (m, 3))
    oxy = rng.uniform(0, 1600, (m, 2)).astype(np.float32)
    # hostile tail: points on / behind the camera plane, absurd magnitudes, zeros, NaN; cameras with tiny / huge / zero / NaN entries
    k = m // 10
    X[:k] *= 10.0 ** rng.integers(-40, 40, (k, 1))
    X[k:2 * k, rng.integers(0, 3)] = 0.0
    X[2 * k:2 * k + 50] = np.nan
    X[2 * k + 50:2 * k + 100] = np.inf
    X[2 * k + 100:2 * k + 200] = 0.0
    # (camera entries stay zero or within 2^-100 .. 2^100: that half of the guard is eg3d_create's, checked on the host)
    Pi[3 * k:4 * k] *= (10.0 ** rng.integers(-24, 24, (k, 1))).astype(np.float32)
    Pi[4 * k:4 * k + 100] = 0.0
    nz = Pi[:, :12][Pi[:, :12] != 0]
    assert np.isfinite(Pi).all() and np.abs(nz).min() >= 2.0 ** -100 and np.abs(nz).max() <= 2.0 ** 100
    X = np.ascontiguousarray(X)
    out = np.zeros(16 * m)
    assert c.eg3d_probe_gn_rows(m, D.np_ptr(Pi, C.c_float), D.np_ptr(oxy, C.c_float), D.np_ptr(X, C.c_double), D.np_ptr(out, C.c_double)) == 0
    a, b = out[:8 * m].view(np.uint64), out[8 * m:].view(np.uint64)
    same = (a == b) | (np.isnan(out[:8 * m]) & np.isnan(out[8 * m:]))
    assert same.all(), int((~same).sum())
