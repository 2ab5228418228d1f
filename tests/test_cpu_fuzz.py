"""CPU twin of tests/test_gpu_fuzz.py: on randomised, mutated scenes (tests/fuzz_scenes.py) the DEVICE code of
stage B compiled for the host (tests/hostsim) must equal the oracle bit for bit. Independent of a GPU, so it runs
in every round's CPU suite; the device run of the same cases (all kernels, through the C ABI) is the gpu test."""
import ctypes as C

import pytest

import hostsim_binding as hs
from fuzz_scenes import HOSTILE_KINDS, draw, hostile
from oracle import binding as ob
from parity_util import compare_edgepoints


@pytest.mark.parametrize("case", [0, 1, 2, 3, 5, 6, 9, 13])
def test_random_mutated_scene_hostsim_vs_oracle(case):
    s, sa, seeds = draw(case)
    n = len(seeds.trk_off) - 1
    o = ob.Oracle(C.byref(sa.c))
    ref = o.match(C.byref(seeds.c), 0, n, 1)
    cand = o.candidates_raw(C.byref(seeds.c), 0, n)
    # the team form of the expand stage (plain / slot step by case), then the chain state machine the engine kernel runs
    for mode in (int(bool(case & 1)), 2, 3):
        got = hs.match(C.byref(sa.c), C.byref(seeds.c), 0, n, cand, slot_step=mode)
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (case, mode, rep["msgs"][:3])
        assert got["n_chains"] == ref["stats"]["n_chains"] and got["n_tasks"] == ref["stats"]["n_tasks"]
        assert got["flags"] == ref["flags"], (case, mode)


@pytest.mark.parametrize("case", range(6))
def test_hostile_numeric_inputs_hostsim_vs_oracle(case):
    """NaN / inf / huge seed observations, zero-length segments, a zero camera, a NaN fundamental matrix: the
    device code compiled for the host still equals the oracle (the GPU run of the same cases is the gpu test)."""
    s, sa, seeds = hostile(case, [HOSTILE_KINDS[case]])
    n = len(seeds.trk_off) - 1
    o = ob.Oracle(C.byref(sa.c))
    ref = o.match(C.byref(seeds.c), 0, n, 1)
    cand = o.candidates_raw(C.byref(seeds.c), 0, n)
    for mode in (0, 2, 3):
        got = hs.match(C.byref(sa.c), C.byref(seeds.c), 0, n, cand, slot_step=mode)
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"], (case, mode, rep["msgs"][:3])


def _mutated_cloud(case):
    """An oracle cloud of a mutated scene, then made irregular the way a caller could hand it back: repeated 3-D
    coordinates inside and across chains, NaN coordinates, observations on / outside the image border, NaN
    observation coordinates."""
    import numpy as np
    s, sa, seeds = draw(case)
    n = len(seeds.trk_off) - 1
    o = ob.Oracle(C.byref(sa.c))
    cloud = o.match(C.byref(seeds.c), 0, n, 1)
    rng = np.random.default_rng(500 + case)
    X, xy = cloud["X"].copy(), cloud["obs_xy"].copy()
    m = len(X)
    if m > 20:
        i = rng.integers(1, m, 12)
        X[i] = X[i - 1]                                   # consecutive duplicates: no edge, same node
        j = rng.integers(0, m, 8)
        X[j] = X[rng.integers(0, m, 8)]                   # far duplicates: node shared between chains
        X[rng.integers(0, m, 3)] = np.nan
        k = rng.integers(0, len(xy), 10)
        xy[k[:3]] = [0.0, 5.0]
        xy[k[3:6]] = [float(sa.c.width), float(sa.c.height)]
        xy[k[6:8]] = [-4.0, 1e9]
        xy[k[8:]] = np.nan
    cloud["X"], cloud["obs_xy"] = X, xy
    return sa, o, cloud


@pytest.mark.parametrize("case", [0, 2, 4, 7, 8, 14])
def test_replay_and_dedup_on_irregular_clouds_match_oracle(case):
    """Row a17 replay and the 3 px dedup (N3) on irregular clouds: product (flat tables) == oracle (the reference's
    containers), including NaN coordinates and observations off the image (the dedup's bounds rule)."""
    import numpy as np
    from edgegraph3d_amd import _cdefs as D
    from edgegraph3d_amd import host
    sa, o, cloud = _mutated_cloud(case)
    got, ref = host.replay_matches(C.byref(sa.c), cloud), o.replay_matches(cloud)
    assert set(got) == set(ref)
    for f in ref:
        x, y = got[f], ref[f]
        if isinstance(y, np.ndarray):
            assert np.array_equal(x.view(np.uint32) if x.dtype.kind == "f" else x, y.view(np.uint32) if y.dtype.kind == "f" else y), (case, f)
        else:
            assert x == y, (case, f)
    ep = D.EdgePointsArrays(cloud)
    npts = int(cloud["n_points"])
    keep_o, keep_h = np.zeros(max(npts, 1), np.uint8), np.zeros(max(npts, 1), np.uint8)
    assert ob.lib().orc_filter_close_2d(o._h, C.byref(ep.c), D.np_ptr(keep_o, C.c_uint8)) == 0
    assert host.lib().eg3d_host_filter_close_2d(sa.c.n_views, sa.c.width, sa.c.height, C.byref(ep.c), D.np_ptr(keep_h, C.c_uint8)) == 0
    assert np.array_equal(keep_o, keep_h), case
