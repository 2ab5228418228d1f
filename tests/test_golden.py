"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle — the reference has none). CPU: the synthetic generator and the oracle still reproduce the
fixture bit-for-bit. GPU: the HIP path reproduces it through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

from edgegraph3d_amd import api, host
from parity_util import compare_edgepoints

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "synthetic_tiny_v1.npz"))
    scene = {k[len("scene_"):]: z[k] for k in z.files if k.startswith("scene_")}
    for k in ("n_views", "width", "height"):
        scene[k] = int(scene[k])
    return z, scene


def expected(z):
    d = {k: z["out_" + k] for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")}
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


def test_oracle_reproduces_fixture_outputs():
    from oracle import binding as ob
    z, scene = load()
    sa = host.SceneArrays(scene)
    se = host.SeedsArrays(z["seeds_trk_off"], z["seeds_trk_view"], z["seeds_trk_xy"])
    o = ob.Oracle(C.byref(sa.c))
    r = o.match(C.byref(se.c), 0, len(z["seeds_trk_off"]) - 1, 1)
    rep = compare_edgepoints(expected(z), r)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep
    Xo, inl = o.gn_filter(z["gn_X"], z["gn_off"], z["gn_view"], z["gn_xy"], 3.0)
    assert np.array_equal(inl, z["gn_inlier"]) and np.array_equal(Xo.view(np.uint32), z["gn_Xout"].view(np.uint32))


@pytest.mark.gpu
def test_hip_path_reproduces_fixture():
    z, scene = load()
    sa = host.SceneArrays(scene)
    se = host.SeedsArrays(z["seeds_trk_off"], z["seeds_trk_view"], z["seeds_trk_xy"])
    ctx = api.Context(C.byref(sa.c))
    got = ctx.match_refpoints(C.byref(se.c), 0, len(z["seeds_trk_off"]) - 1)
    rep = compare_edgepoints(expected(z), got, rel_tol=1e-4)
    assert rep["ok"], rep["msgs"]
    c = ctx.candidates(C.byref(se.c), 0, len(z["seeds_trk_off"]) - 1)
    for k in ("cand_off", "cand_pl", "start_off", "start_pl", "start_seg", "task_sv", "task_hit", "task_list_off",
              "list_off", "hit_pl", "hit_seg"):
        assert np.array_equal(c[k], z["cand_" + k]), k
    Xo, inl, _ = ctx.gn_filter(z["gn_X"], z["gn_off"], z["gn_view"], z["gn_xy"], 3.0)
    assert np.array_equal(inl, z["gn_inlier"]) and np.array_equal(Xo.view(np.uint32), z["gn_Xout"].view(np.uint32))
    ctx.close()
