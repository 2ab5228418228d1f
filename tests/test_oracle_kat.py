"""Hand-derived known-answer tests for the primitives of the path. The reference has
no tests or golden vectors (parity unpinned), so these KATs — whose answers follow from the maths
and from the reference code's stated semantics — are what pins the oracle. Each KAT is also run
through the device-side body (host instantiation, tests/hostsim) where one exists."""
import ctypes as C
import math

import numpy as np
import pytest

from edgegraph3d_amd import _cdefs as D
from oracle import binding as ob
import hostsim_binding as hs


def f32(*v):
    return np.array(v, dtype=np.float32)


@pytest.fixture(scope="module")
def L():
    return ob.lib()


def test_squared_distance_rounds_once_from_double(L):
    # (3,4) -> 25 exactly; tiny differences must not be flushed by float squaring
    assert L.orc_squared_2d_distance(0, 0, 3, 4) == 25.0
    a, b = np.float32(1.0), np.float32(1.0) + np.float32(2 ** -23)
    d = (np.float64(np.float32(a - b))) ** 2
    assert L.orc_squared_2d_distance(a, 0, b, 0) == np.float32(d)
    assert hs.lib().hostsim_dist2(a, 0, b, 0) == np.float32(d)


def test_point_segment_distance_clamps_and_degenerate(L):
    proj = f32(0, 0)
    # projection inside the segment
    d = L.orc_minimum_distancesq(5, 3, 0, 0, 10, 0, D.np_ptr(proj, C.c_float))
    assert d == 9.0 and tuple(proj) == (5.0, 0.0)
    # before the start -> clamps to v
    d = L.orc_minimum_distancesq(-3, 4, 0, 0, 10, 0, D.np_ptr(proj, C.c_float))
    assert d == 25.0 and tuple(proj) == (0.0, 0.0)
    # beyond the end -> clamps to w
    d = L.orc_minimum_distancesq(13, 4, 0, 0, 10, 0, D.np_ptr(proj, C.c_float))
    assert d == 25.0 and tuple(proj) == (10.0, 0.0)
    # degenerate segment v == w
    d = L.orc_minimum_distancesq(3, 4, 1, 1, 1, 1, D.np_ptr(proj, C.c_float))
    assert d == 13.0 and tuple(proj) == (1.0, 1.0)
    q = f32(0, 0)
    assert hs.lib().hostsim_seg_closest(13, 4, 0, 0, 10, 0, D.np_ptr(q, C.c_float)) == 25.0 and tuple(q) == (10.0, 0.0)


def test_point_segment_distances_are_never_negative_so_their_bits_order_like_their_values(L):
    """K1 picks a candidate polyline's first closest segment as the minimum of 64-bit keys (bits of the squared distance :
    segment index) in LDS. That is the oracle's `d < best` scan (smallest distance, then smallest index) iff the squared
    distances are >= +0 — never -0, never negative — because only then do IEEE-754 bit patterns order like the values.
    minimum_distancesq returns dx*dx + dy*dy of single-precision differences: checked here on random and edge inputs
    (zero-length segments, the point on the segment, on a vertex, huge and tiny coordinates), on the oracle and on the
    device-side body."""
    rng = np.random.default_rng(20240929)
    n = 200000
    scale = rng.choice(np.float32([1e-30, 1e-3, 1.0, 1e3, 1e7]), n)
    P = (rng.normal(0, 1, (n, 6)).astype(np.float32) * scale[:, None]).astype(np.float32)
    P[::7, 4:6] = P[::7, 2:4]          # zero-length segments
    P[::11, 0:2] = P[::11, 2:4]        # the point on the first vertex
    P[::13, 0:2] = (P[::13, 2:4] + P[::13, 4:6]) * np.float32(0.5)   # (about) on the segment
    proj = f32(0, 0)
    H = hs.lib()
    d_or = np.array([L.orc_minimum_distancesq(*map(float, r), D.np_ptr(proj, C.c_float)) for r in P[:20000]], np.float32)
    d_dev = np.array([H.hostsim_seg_closest(*map(float, r), D.np_ptr(proj, C.c_float)) for r in P[:20000]], np.float32)
    assert np.array_equal(d_or.view(np.uint32), d_dev.view(np.uint32))
    assert not np.signbit(d_or).any() and not np.isnan(d_or).any()
    # the ordering claim itself, on all non-negative single-precision values incl. +0, subnormals and +inf
    a = np.abs(rng.normal(0, 1, n).astype(np.float32) * scale)
    a[:4] = [0.0, np.float32(1e-45), np.finfo(np.float32).max, np.inf]
    b = np.roll(a, 1)
    assert np.array_equal(a < b, a.view(np.uint32) < b.view(np.uint32))
    assert np.array_equal(a == b, a.view(np.uint32) == b.view(np.uint32))


def test_segment_line_intersection_endpoints_and_parallel(L):
    inter = f32(0, 0)
    par, ovl = C.c_int(), C.c_int()
    line = f32(1, 0, -5)  # x = 5
    # crossing in the middle: t = 0.5
    assert L.orc_intersect_segment_line(0, 0, 10, 2, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(par), C.byref(ovl)) == 1
    assert tuple(inter) == (5.0, 1.0)
    # t == 0 and t == 1 are hits (closed interval)
    assert L.orc_intersect_segment_line(5, 0, 10, 0, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(par), C.byref(ovl)) == 1
    assert L.orc_intersect_segment_line(0, 0, 5, 7, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(par), C.byref(ovl)) == 1
    assert tuple(inter) == (5.0, 7.0)
    # just outside
    assert L.orc_intersect_segment_line(0, 0, 4.99, 0, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(par), C.byref(ovl)) == 0
    # parallel, not overlapped / overlapped
    assert L.orc_intersect_segment_line(3, 0, 3, 9, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(par), C.byref(ovl)) == 0
    assert par.value == 1 and ovl.value == 0
    assert L.orc_intersect_segment_line(5, 0, 5, 9, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(par), C.byref(ovl)) == 0
    assert par.value == 1 and ovl.value == 1


def test_quasi_parallel_guard_is_signed_and_distance_limited(L):
    """Q14: cos between the segment in walking order and the line direction (1, -a/b); only
    cos > 0.965 within 5 px is flagged; an anti-parallel segment is not."""
    inter = f32(0, 0)
    qp, dist = C.c_int(), C.c_float()
    line = f32(0.05, 1.0, -1.0)  # almost horizontal: y = 1 - 0.05 x ; direction (1, -0.05)
    n = math.hypot(0.05, 1.0)
    line = f32(0.05 / n, 1.0 / n, -1.0 / n)
    # segment walking +x, slightly below the line, 2 px away -> quasi-parallel within distance
    L.orc_intersect_segment_line_nqp(0, -1, 10, -1.4, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(qp), C.byref(dist))
    assert qp.value == 1 and 0 < dist.value <= 5
    # same segment walked the other way (anti-parallel): not flagged
    L.orc_intersect_segment_line_nqp(10, -1.4, 0, -1, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(qp), C.byref(dist))
    assert qp.value == 0
    # parallel-ish but 20 px away: not flagged
    L.orc_intersect_segment_line_nqp(0, -20, 10, -20.4, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(qp), C.byref(dist))
    assert qp.value == 0
    # a crossing at a healthy angle is found and not flagged
    found = L.orc_intersect_segment_line_nqp(5, -5, 5, 5, D.np_ptr(line, C.c_float), D.np_ptr(inter, C.c_float), C.byref(qp), C.byref(dist))
    assert found == 1 and qp.value == 0


def test_cell_rounding_at_multiples_of_cell_size(L):
    col, row, br, bc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    L.orc_cell_from_coords(30.0, 45.0, 75.0, C.byref(col), C.byref(row), C.byref(br), C.byref(bc))
    assert (col.value, row.value, br.value, bc.value) == (1, 2, 0, 0)
    # exact multiple of the cell size in x: boundary flag, index rounds to that multiple
    L.orc_cell_from_coords(30.0, 60.0, 75.0, C.byref(col), C.byref(row), C.byref(br), C.byref(bc))
    assert (col.value, br.value, bc.value) == (2, 1, 0)
    # within 1e-3 below a multiple: treated as on the boundary and rounded UP
    L.orc_cell_from_coords(4.0, 7.9999, 2.0, C.byref(col), C.byref(row), C.byref(br), C.byref(bc))
    assert (col.value, br.value) == (2, 1)
    L.orc_cell_from_coords(4.0, 7.9, 2.0, C.byref(col), C.byref(row), C.byref(br), C.byref(bc))
    assert (col.value, br.value) == (1, 0)


def test_epiline_is_normalised_in_double(L):
    F = np.array([0, 0, 3, 0, 0, 4, 1, 2, 5], dtype=np.float64)
    line = f32(0, 0, 0)
    assert L.orc_epiline(D.np_ptr(F, C.c_double), 10.0, 20.0, D.np_ptr(line, C.c_float)) == 1
    # l = (3, 4, 10+40+5) / 5
    assert np.allclose(line, [0.6, 0.8, 11.0], rtol=0, atol=1e-7)
    assert line[0] == np.float32(3 / 5) and line[2] == np.float32(55 / 5)


def _cams():
    """Three simple cameras: f=1000, pp=(500,400), looking down +Z from x = -100, 0, +100."""
    P = np.zeros((3, 16), np.float32)
    for i, cx in enumerate((-100.0, 0.0, 100.0)):
        K = np.array([[1000, 0, 500], [0, 1000, 400], [0, 0, 1]], np.float64)
        Rt = np.hstack([np.eye(3), np.array([[-cx], [0], [0]])])
        P[i, :12] = (K @ Rt).astype(np.float32).reshape(-1)
    return P


def _proj(P, X):
    h = P[:12].reshape(3, 4).astype(np.float64) @ np.append(X, 1.0)
    return (h[0] / h[2], h[1] / h[2])


def test_triangulation_of_exact_projections(L):
    P = _cams()
    X = np.array([20.0, -30.0, 1000.0])
    xy = np.array([_proj(P[i], X) for i in range(3)], np.float32)
    ids = (C.c_int * 3)(0, 1, 2)
    out = f32(0, 0, 0)
    deg = C.c_int()
    assert L.orc_triangulate(D.np_ptr(P, C.c_float), ids, D.np_ptr(xy, C.c_float), 3, D.np_ptr(out, C.c_float), C.byref(deg)) == 1
    assert deg.value == 0
    assert np.allclose(out, X, rtol=1e-5)
    Xd = f32(0, 0, 0)
    views = np.array([0, 1, 2], np.int32)
    assert hs.lib().hostsim_triangulate(D.np_ptr(P, C.c_float), D.np_ptr(views, C.c_int32), D.np_ptr(xy, C.c_float), 3, D.np_ptr(Xd, C.c_float)) == 1
    assert np.array_equal(Xd.view(np.uint32), out.view(np.uint32)), "device body != oracle bit-for-bit"


def test_triangulation_rejects_large_residual(L):
    """mse >= 9 px^2 -> invalid (triangulation.cpp:168)."""
    P = _cams()
    X = np.array([20.0, -30.0, 1000.0])
    xy = np.array([_proj(P[i], X) for i in range(3)], np.float32)
    xy[1, 1] += 30.0  # 30 px vertical outlier in the middle view
    ids = (C.c_int * 3)(0, 1, 2)
    out = f32(0, 0, 0)
    deg = C.c_int()
    assert L.orc_triangulate(D.np_ptr(P, C.c_float), ids, D.np_ptr(xy, C.c_float), 3, D.np_ptr(out, C.c_float), C.byref(deg)) == 0


def test_gn_fails_on_degenerate_geometry(L):
    """det(H) < 1e-5 -> invalid: a single far point seen by two cameras at the same position."""
    P = _cams()
    P2 = np.stack([P[1], P[1], P[1]])
    X = np.array([0.0, 0.0, 1e6])
    xy = np.array([_proj(P2[i], X) for i in range(3)], np.float32)
    xy[0, 0] += 0.5
    ids = (C.c_int * 3)(0, 1, 2)
    out = f32(0, 0, 0)
    deg = C.c_int()
    assert L.orc_triangulate(D.np_ptr(P2, C.c_float), ids, D.np_ptr(xy, C.c_float), 3, D.np_ptr(out, C.c_float), C.byref(deg)) == 0


def _polyline():
    # an L-shaped polyline: (0,0)-(30,0)-(30,40); node ids start=7, end=9
    return f32(0, 0, 30, 0, 30, 40), 3, 7, 9


def test_next_point_by_distance_across_vertices_both_directions(L):
    v, n, s, e = _polyline()
    oseg, oxy = C.c_uint32(), f32(0, 0)
    # from (10,0) on segment 0 towards the end, 10 px: stays on segment 0
    assert L.orc_next_by_distance(D.np_ptr(v, C.c_float), n, s, e, 0, 10, 0, e, 10.0, C.byref(oseg), D.np_ptr(oxy, C.c_float)) == 0
    assert oseg.value == 0 and tuple(oxy) == (20.0, 0.0)
    # from (25,0), 10 px (Euclidean from the start point): crosses the vertex (30,0)
    assert L.orc_next_by_distance(D.np_ptr(v, C.c_float), n, s, e, 0, 25, 0, e, 10.0, C.byref(oseg), D.np_ptr(oxy, C.c_float)) == 0
    assert oseg.value == 1 and oxy[0] == 30.0 and 5.0 < oxy[1] < 10.0
    # reaching the extreme returns the end point and the flag
    assert L.orc_next_by_distance(D.np_ptr(v, C.c_float), n, s, e, 1, 30, 35, e, 10.0, C.byref(oseg), D.np_ptr(oxy, C.c_float)) == 1
    assert oseg.value == 1 and tuple(oxy) == (30.0, 40.0)
    # towards the start from (30,5) on segment 1
    assert L.orc_next_by_distance(D.np_ptr(v, C.c_float), n, s, e, 1, 30, 5, s, 10.0, C.byref(oseg), D.np_ptr(oxy, C.c_float)) == 0
    assert oseg.value == 0 and oxy[1] == 0.0 and 20.0 < oxy[0] < 25.0
    hseg, hxy = C.c_uint32(), f32(0, 0)
    w = hs.lib().hostsim_walk_by_distance(D.np_ptr(v, C.c_float), n, s, e, 1, 30, 5, s, 10.0, C.byref(hseg), D.np_ptr(hxy, C.c_float))
    assert w == 1 and hseg.value == oseg.value and np.array_equal(hxy.view(np.uint32), oxy.view(np.uint32))


def test_next_point_by_line_bounded_window(L):
    v, n, s, e = _polyline()
    oseg, oxy, flags = C.c_uint32(), f32(0, 0), C.c_int()
    line = f32(1, 0, -18)  # x = 18
    # from (10,0) towards the end: hit at (18,0), 8 px away: inside [5,20]
    assert L.orc_next_by_line(D.np_ptr(v, C.c_float), n, s, e, 0, 10, 0, e, D.np_ptr(line, C.c_float), 1, 5.0, 20.0, C.byref(oseg), D.np_ptr(oxy, C.c_float), C.byref(flags)) == 1
    assert oseg.value == 0 and tuple(oxy) == (18.0, 0.0)
    # from (15,0): hit 3 px away: bounded-distance violation
    assert L.orc_next_by_line(D.np_ptr(v, C.c_float), n, s, e, 0, 15, 0, e, D.np_ptr(line, C.c_float), 1, 5.0, 20.0, C.byref(oseg), D.np_ptr(oxy, C.c_float), C.byref(flags)) == 0
    assert flags.value & 4
    # unbounded form accepts it
    assert L.orc_next_by_line(D.np_ptr(v, C.c_float), n, s, e, 0, 15, 0, e, D.np_ptr(line, C.c_float), 0, 0, 0, C.byref(oseg), D.np_ptr(oxy, C.c_float), C.byref(flags)) == 1
    # walking towards the start from (25,0) never meets x=28 -> extreme
    line2 = f32(1, 0, -28)
    assert L.orc_next_by_line(D.np_ptr(v, C.c_float), n, s, e, 0, 25, 0, s, D.np_ptr(line2, C.c_float), 0, 0, 0, C.byref(oseg), D.np_ptr(oxy, C.c_float), C.byref(flags)) == 0
    assert flags.value & 2
    # a line almost along segment 1 (x = 30.5, vertical) stops the walk as quasi-parallel
    line3 = f32(1, 0, -30.5)
    assert L.orc_next_by_line(D.np_ptr(v, C.c_float), n, s, e, 0, 25, 0, e, D.np_ptr(line3, C.c_float), 0, 0, 0, C.byref(oseg), D.np_ptr(oxy, C.c_float), C.byref(flags)) == 0
    assert flags.value & 1
    hseg, hxy = C.c_uint32(), f32(0, 0)
    w = hs.lib().hostsim_walk_by_line(D.np_ptr(v, C.c_float), n, s, e, 0, 25, 0, e, D.np_ptr(line3, C.c_float), 0, 0, 0, C.byref(hseg), D.np_ptr(hxy, C.c_float))
    assert w == 4  # WALK_QUASIPARALLEL
