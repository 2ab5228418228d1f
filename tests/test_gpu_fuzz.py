"""Randomised scenes, device vs oracle: the committed configs exercise a handful of shapes; this draws
small scenes with random rigs, noise levels and image sizes and then MUTATES them the way real inputs
are irregular — view pairs without a fundamental matrix, loop polylines (start == end, Q8), invalid
polylines that kept their vertices, tracks that repeat a view id (Q2), observations on or outside the
image border (Q7), very short tracks — and requires the full path, stage A and the polyline-set path to
agree with the oracle bit for bit on every one. Seeds are fixed: a failure names its case."""
import ctypes as C

import numpy as np
import pytest

from edgegraph3d_amd import api, host
from fuzz_scenes import HOSTILE_KINDS, draw, hostile
from parity_util import compare_edgepoints

pytestmark = pytest.mark.gpu

CASES = list(range(40)) + [100, 101, 102]


def _oracle(scene_ptr):
    from oracle import binding as ob
    return ob.Oracle(scene_ptr)


@pytest.mark.parametrize("case", CASES)
def test_random_mutated_scene_matches_oracle(case):
    s, sa, seeds = draw(case)
    n = len(seeds.trk_off) - 1
    ctx = api.Context(C.byref(sa.c))
    orc = _oracle(C.byref(sa.c))
    got = ctx.match_refpoints(C.byref(seeds.c), 0, n)
    ref = orc.match(C.byref(seeds.c), 0, n, nthreads=8)
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (case, rep["msgs"][:3])
    assert got["flags"] == ref["flags"], (case, got["flags"], ref["flags"])
    # stage A alone
    ca, cb = ctx.candidates(C.byref(seeds.c), 0, n), orc.candidates(C.byref(seeds.c), 0, n)
    for k in cb:
        x, y = ca[k], cb[k]
        if isinstance(y, np.ndarray):
            bits = (lambda a: a.view(np.uint32) if a.dtype == np.float32 else a)
            assert np.array_equal(bits(x), bits(y)), (case, k)
        else:
            assert x == y, (case, k, x, y)
    if case >= 100:   # many views: the polyline-set oracle alone takes minutes there
        ctx.close()
        return
    # the polyline-set path on the same mutated scene
    n_sets, row_off, ids = s.polyline_sets(2)
    gs = ctx.match_polyline_sets(n_sets, row_off, ids)
    rs = orc.match_polyline_sets(n_sets, row_off, ids, 0, n_sets, 8)
    rep = compare_edgepoints(rs, gs)
    assert rep["ok"] and rep["bitexact_X"], (case, rep["msgs"][:3])
    ctx.close()


@pytest.mark.parametrize("case", range(6))
def test_gn_filter_degenerate_inputs_match_oracle(case):
    """Config 5 on inputs a real SfM file can hold: NaN / huge / zero coordinates, points behind the cameras,
    gross outliers, all observations from one view (singular normal equations), tracks of 0, 1 and 2
    observations — both abs() behaviours (Q9). X and the inlier flags must equal the oracle's bit for bit."""
    from oracle import binding as ob
    s = host.Synth(5)
    ctx = api.Context(s.scene)
    orc = ob.Oracle(s.scene)
    rng = np.random.default_rng(77 + case)
    X, off, view, xy = s.points(3000 + case)
    X, xy, view = X.copy(), xy.copy(), view.copy()
    n = len(X)
    idx = rng.integers(0, n, 60)
    X[idx[:10]] = np.nan
    X[idx[10:20]] *= 1e6
    X[idx[20:30]] = 0
    X[idx[30:40]] = -X[idx[30:40]]
    for p in idx[40:50]:
        xy[off[p]:off[p + 1]] += rng.normal(0, 200, (off[p + 1] - off[p], 2)).astype(np.float32)
    for p in idx[50:60]:
        view[off[p]:off[p + 1]] = view[off[p]]
    keep = np.ones(len(view), bool)
    for j, p in enumerate(rng.integers(0, n, 45)):
        a, b = int(off[p]), int(off[p + 1])
        keep[a + (j % 3):b] = False
    noff = np.zeros_like(off)
    noff[1:] = np.cumsum([keep[off[p]:off[p + 1]].sum() for p in range(n)])
    view2, xy2 = view[keep], xy[keep]
    for legacy in (False, True):
        Xo, inl = orc.gn_filter(X, noff, view2, xy2, 3.0, legacy_abs=legacy)
        Xg, ing, _ = ctx.gn_filter(X, noff, view2, xy2, 3.0, legacy_abs=legacy)
        diff = np.nonzero((Xo.view(np.uint32) != Xg.view(np.uint32)).any(1) | (inl != ing))[0]
        assert len(diff) == 0, (case, legacy, [(int(noff[p + 1] - noff[p]), Xo[p], Xg[p], inl[p], ing[p]) for p in diff[:3]])
    ctx.close()


@pytest.mark.parametrize("case", range(12))
def test_hostile_numeric_inputs_match_oracle(case):
    """NaN / inf / 1e12 seed observations, zero-length segments, a camera matrix of zeros, a NaN fundamental
    matrix — alone (cases 0-5) and three at a time: inputs on which conversions and comparisons could differ between
    x86 and the GPU. The result must still equal the oracle's bit for bit."""
    ok_kinds = HOSTILE_KINDS[:6]
    kinds = [ok_kinds[case]] if case < 6 else list(np.random.default_rng(case).choice(ok_kinds, 3, replace=False))
    s, sa, seeds = hostile(case, kinds)
    n = len(seeds.trk_off) - 1
    ctx = api.Context(C.byref(sa.c))
    got = ctx.match_refpoints(C.byref(seeds.c), 0, n)
    ref = _oracle(C.byref(sa.c)).match(C.byref(seeds.c), 0, n, nthreads=8)
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (case, kinds, rep["msgs"][:3])
    ctx.close()


@pytest.mark.parametrize("kind", ["vtx_nan", "vtx_huge"])
def test_scene_with_non_finite_or_absurd_vertices_is_refused(kind):
    """include/eg3d.h: vertices of valid polylines must be finite and within +-1e7 px — the reference's grid
    sampling walks every segment in ~2.6 px steps wherever it lies, so a 1e20 coordinate never finishes."""
    s, sa, seeds = hostile(4, [kind])
    with pytest.raises(RuntimeError, match="not finite or beyond"):
        api.Context(C.byref(sa.c))


def test_one_context_many_calls_of_different_shapes(monkeypatch):
    """A context's buffers only grow and are reused by every call; device-only calls accumulate their cloud over
    chunks. A random sequence of calls on ONE context — seed ranges of very different sizes, host-copy and
    device-only calls alternating, forced chunking — must give, call by call, what a fresh computation gives."""
    monkeypatch.setenv("EG3D_MAX_SCRATCH_MB", "64")      # several chunks on the larger ranges
    s = host.Synth(2)
    ctx = api.Context(s.scene)
    monkeypatch.delenv("EG3D_MAX_SCRATCH_MB")
    ctx.upload_seeds(s.seeds)
    whole = ctx.match_resident(0, s.n_seeds)
    key0 = whole["key"][:, 0]
    rng = np.random.default_rng(31)

    def expect(b, e):
        sel = (key0 >= b) & (key0 < e)
        first = int(np.argmax(sel)) if sel.any() else 0
        n = int(sel.sum())
        o0, o1 = int(whole["obs_off"][first]), int(whole["obs_off"][first + n])
        return {"X": whole["X"][first:first + n], "key": whole["key"][first:first + n],
                "obs_off": whole["obs_off"][first:first + n + 1] - o0, "obs_view": whole["obs_view"][o0:o1],
                "obs_pl": whole["obs_pl"][o0:o1], "obs_seg": whole["obs_seg"][o0:o1], "obs_xy": whole["obs_xy"][o0:o1]}

    for it in range(14):
        b = int(rng.integers(0, s.n_seeds))
        e = int(min(s.n_seeds, b + rng.choice([0, 1, 7, 60, 400, 2000])))
        want = expect(b, e)
        if it % 2:
            ctx.match_resident(b, e, device_only=True)
            got = ctx.fetch_device_output()
        else:
            got = ctx.match_resident(b, e)
        for k in ("X", "obs_xy"):
            assert np.array_equal(got[k].view(np.uint32).ravel(), want[k].view(np.uint32).ravel()), (it, b, e, k)
        for k in ("key", "obs_off", "obs_view", "obs_pl", "obs_seg"):
            assert np.array_equal(got[k], want[k]), (it, b, e, k)
    ctx.close()


def test_host_copy_of_a_large_cloud_equals_the_device_arrays():
    """Independent of the oracle: the cloud a host-copy call returns must be, byte for byte, what a device-only call
    leaves in HBM (fetched with plain hipMemcpy) — on a cloud large enough for the staged, multi-threaded copy
    (0.45 GB) and on odd sub-ranges whose array sizes vary."""
    s = host.Synth(3)
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    for b, e in ((0, s.n_seeds), (17, 4001), (1234, 5555), (3000, 6268)):
        hostc = ctx.match_resident(b, e)
        ctx.match_resident(b, e, device_only=True)
        dev = ctx.fetch_device_output()
        for k in ("X", "obs_xy"):
            assert np.array_equal(hostc[k].view(np.uint32).ravel(), dev[k].view(np.uint32).ravel()), (b, e, k)
        for k in ("key", "obs_off", "obs_view", "obs_pl", "obs_seg"):
            assert np.array_equal(hostc[k], dev[k]), (b, e, k)
    ctx.close()
