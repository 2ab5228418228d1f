"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle — the reference has none). CPU: the synthetic generator and the oracle still reproduce the
fixture bit-for-bit. GPU: the HIP path reproduces it through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import forms
from edgegraph3d_amd import api, host
from parity_util import compare_edgepoints

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "synthetic_tiny_v1.npz"))
    scene = {k[len("scene_"):]: z[k] for k in z.files if k.startswith("scene_")}
    for k in ("n_views", "width", "height"):
        scene[k] = int(scene[k])
    return z, scene


def expected(z, rows):
    """The fixture's outputs for one DLT form (the inputs, stage A and the filter part do not depend on it)."""
    zo = z if rows == 2 else np.load(forms.golden_path("synthetic_tiny_v1", rows))
    d = {k: zo["out_" + k] for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")}
    d["n_points"], d["n_obs"] = len(d["X"]), len(d["obs_view"])
    return d


def test_generator_reproduces_fixture_inputs():
    z, scene = load()
    s = host.Synth(0)
    sc = s.scene_np()
    for k, v in sc.items():
        assert np.array_equal(np.asarray(v), np.asarray(scene[k])), k
    off, view, xy = s.seeds_np()
    assert np.array_equal(off, z["seeds_trk_off"]) and np.array_equal(view, z["seeds_trk_view"])
    assert np.array_equal(xy.view(np.uint32), z["seeds_trk_xy"].view(np.uint32))


@pytest.mark.parametrize("rows", forms.FORMS, ids=[forms.IDS[r] for r in forms.FORMS])
def test_oracle_reproduces_fixture_outputs(rows):
    from oracle import binding as ob
    z, scene = load()
    sa = host.SceneArrays(scene)
    se = host.SeedsArrays(z["seeds_trk_off"], z["seeds_trk_view"], z["seeds_trk_xy"])
    o = ob.Oracle(C.byref(sa.c))
    with forms.oracle_rows(rows):
        r = o.match(C.byref(se.c), 0, len(z["seeds_trk_off"]) - 1, 1)
    rep = compare_edgepoints(expected(z, rows), r)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep
    Xo, inl = o.gn_filter(z["gn_X"], z["gn_off"], z["gn_view"], z["gn_xy"], 3.0)
    assert np.array_equal(inl, z["gn_inlier"]) and np.array_equal(Xo.view(np.uint32), z["gn_Xout"].view(np.uint32))


@pytest.mark.gpu
def test_hip_path_reproduces_fixture(eg3d_form):
    z, scene = load()
    sa = host.SceneArrays(scene)
    se = host.SeedsArrays(z["seeds_trk_off"], z["seeds_trk_view"], z["seeds_trk_xy"])
    ctx = api.Context(C.byref(sa.c))
    got = ctx.match_refpoints(C.byref(se.c), 0, len(z["seeds_trk_off"]) - 1)
    rep = compare_edgepoints(expected(z, eg3d_form), got, rel_tol=1e-4)
    assert rep["ok"], rep["msgs"]
    c = ctx.candidates(C.byref(se.c), 0, len(z["seeds_trk_off"]) - 1)
    for k in ("cand_off", "cand_pl", "start_off", "start_pl", "start_seg", "task_sv", "task_hit", "task_list_off",
              "list_off", "hit_pl", "hit_seg"):
        assert np.array_equal(c[k], z["cand_" + k]), k
    Xo, inl, _ = ctx.gn_filter(z["gn_X"], z["gn_off"], z["gn_view"], z["gn_xy"], 3.0)
    assert np.array_equal(inl, z["gn_inlier"]) and np.array_equal(Xo.view(np.uint32), z["gn_Xout"].view(np.uint32))
    ctx.close()


# ---- pipelines 1-2 extractor ----
def load_sets(rows=2):
    """The sets and, for one DLT form, the expected output (the sets themselves are in the base file)."""
    z = np.load(os.path.join(HERE, "golden", "synthetic_tiny_sets_v1.npz"))
    if rows != 2:
        zo = np.load(forms.golden_path("synthetic_tiny_sets_v1", rows))
        z = {**{k: z[k] for k in z.files}, **{k: zo[k] for k in zo.files}}
    d = {k: z["out_" + k] for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")}
    d["n_points"], d["n_obs"] = len(d["X"]), len(d["obs_view"])
    return z, d


@pytest.mark.parametrize("rows", forms.FORMS, ids=[forms.IDS[r] for r in forms.FORMS])
def test_oracle_reproduces_sets_fixture(rows):
    from oracle import binding as ob
    z, want = load_sets(rows)
    _, scene = load()
    sa = host.SceneArrays(scene)
    with forms.oracle_rows(rows):
        r = ob.Oracle(C.byref(sa.c)).match_polyline_sets(int(z["n_sets"]), z["row_off"], z["pl_ids"], nthreads=2)
    rep = compare_edgepoints(want, r)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep
    assert [r["stats"]["n_tasks"], r["stats"]["n_chains"], r["flags"]] == list(z["out_counts"])


def test_sets_fixture_invariants():
    """What the extractor promises independently of the oracle's arithmetic: every chain carries its
    sample (key[0]) in the start view (key[1]); consecutive samples of a polyline are 20 px apart
    (Euclidean, next_pl_point_by_distance); every observation lies on a polyline of the point's set."""
    z, d = load_sets()
    _, scene = load()
    V = scene["n_views"]
    key, off = d["key"], d["obs_off"]
    assert (np.diff(key[:, 0].astype(np.int64)) >= 0).all()          # emitted in sample order
    assert (key[:, 1] < V).all() and (key[:, 2] == 0).all()
    # observations only on polylines that belong to some set row of their view
    allowed = [set() for _ in range(V)]
    for r in range(int(z["n_sets"]) * V):
        allowed[r % V].update(int(i) for i in z["pl_ids"][z["row_off"][r]:z["row_off"][r + 1]])
    # points of the three seed views come from the set; expand-all-views may add other polylines
    first = key[:, 3] == 0
    assert first.sum() == int(z["out_counts"][1])                     # one first point per chain


@pytest.mark.gpu
def test_hip_path_reproduces_sets_fixture(eg3d_form):
    z, want = load_sets(eg3d_form)
    _, scene = load()
    sa = host.SceneArrays(scene)
    ctx = api.Context(C.byref(sa.c))
    got = ctx.match_polyline_sets(int(z["n_sets"]), z["row_off"], z["pl_ids"])
    rep = compare_edgepoints(want, got)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep
    ctx.close()
