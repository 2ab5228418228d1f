"""-m gpu: ONE eg3d_match_* call cut into units that run concurrently on the context's internal lanes
(eg3d_set_pipelining, eg3d_api.hip run_pipelined). The reference's parallel entry point runs the seeds of a call on its
OpenMP team and appends in seed order (plg_matching_from_refpoints.cpp:83-104); the property checked here is the same:
whatever the cutting and the number of lanes, the call's cloud is BYTE FOR BYTE the cloud of the uncut call (and that
one is compared with the oracle), in the host arrays and in the device view alike."""
import ctypes as C

import numpy as np
import pytest

from edgegraph3d_amd import api, host
from parity_util import compare_edgepoints

pytestmark = pytest.mark.gpu

ARRAYS = ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")
COUNTS = ("n_points", "n_obs", "n_tasks", "n_hypotheses", "n_chains", "flags")


def _oracle(scene):
    from oracle import binding as ob
    return ob.Oracle(scene)


@pytest.fixture(scope="module")
def have_gpu():
    assert api.device_count() >= 1, "no HIP device: the product path has no CPU fallback"


def _same(a, b, what):
    for k in COUNTS:
        assert a[k] == b[k], (what, k, a[k], b[k])
    for k in ARRAYS:
        x, y = np.ascontiguousarray(a[k]), np.ascontiguousarray(b[k])
        assert x.shape == y.shape and x.dtype == y.dtype, (what, k)
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), (what, k)
    assert a["times"]["bytes_algorithmic"] == b["times"]["bytes_algorithmic"], what


@pytest.mark.parametrize("cfg", [1, 2])
def test_pipelined_call_is_byte_identical_to_the_uncut_call(have_gpu, cfg):
    """lanes x units in {1..4} x {auto, 2, 7, 13}: host arrays, and the device view of a device-only call."""
    s = host.Synth(cfg)
    n = s.n_seeds
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    ctx.set_pipelining(1, 0)
    base = ctx.match_resident(0, n)
    ref = _oracle(s.scene).match(s.seeds, 0, n, nthreads=8)
    rep = compare_edgepoints(ref, base)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"][:3]
    for lanes, units in ((2, 0), (4, 0), (4, 7), (3, 13), (1, 5), (0, 0)):
        ctx.set_pipelining(lanes, units)
        got = ctx.match_resident(0, n)
        _same(base, got, ("host", lanes, units))
        d = ctx.match_resident(0, n, device_only=True)
        dev = ctx.fetch_device_output()
        for k in ("n_points", "n_obs"):
            assert d[k] == base[k]
        for k in ARRAYS:
            assert np.array_equal(np.ascontiguousarray(dev[k]).view(np.uint8), np.ascontiguousarray(base[k]).view(np.uint8)), \
                ("device", lanes, units, k)
    # a sub-range that does not start at seed 0, and an empty one
    ctx.set_pipelining(1, 0)
    sub = ctx.match_resident(n // 5, n - 3)
    ctx.set_pipelining(4, 6)
    _same(sub, ctx.match_resident(n // 5, n - 3), "sub-range")
    e = ctx.match_resident(7, 7)
    assert e["n_points"] == 0 and e["n_obs"] == 0 and len(e["obs_off"]) == 1
    ctx.close()


def test_pipelined_polyline_sets_call_is_byte_identical(have_gpu):
    """The pipelines 1-2 extractor: units are runs of whole sets; key[0] stays the sample index of the CALL."""
    s = host.Synth(1)
    n_sets, row_off, ids = s.polyline_sets()
    ctx = api.Context(s.scene)
    ctx.set_pipelining(1, 0)
    base = ctx.match_polyline_sets(n_sets, row_off, ids)
    ref = _oracle(s.scene).match_polyline_sets(n_sets, row_off, ids, nthreads=8)
    rep = compare_edgepoints(ref, base)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"][:3]
    assert n_sets >= 4
    for lanes, units in ((4, 0), (2, 5), (3, n_sets)):
        ctx.set_pipelining(lanes, units)
        _same(base, ctx.match_polyline_sets(n_sets, row_off, ids), ("sets", lanes, units))
        d = ctx.match_polyline_sets(n_sets, row_off, ids, device_only=True)
        dev = ctx.fetch_device_output()
        assert d["n_points"] == base["n_points"]
        for k in ARRAYS:
            assert np.array_equal(np.ascontiguousarray(dev[k]).view(np.uint8), np.ascontiguousarray(base[k]).view(np.uint8)), \
                ("sets device", lanes, units, k)
    ctx.close()


def test_pipelined_units_cut_into_chunks_and_growing_capacities(have_gpu, monkeypatch):
    """Units whose expand stage is itself cut into several launches (EG3D_MAX_SCRATCH_MB) take their turn once per
    chunk; lanes that start with a tiny hypothesis arena redo their attempt. Still byte-identical."""
    s = host.Synth(1)
    n = s.n_seeds
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    ctx.set_pipelining(1, 0)
    base = ctx.match_resident(0, n)
    ctx.close()
    monkeypatch.setenv("EG3D_MAX_SCRATCH_MB", "4")
    monkeypatch.setenv("EG3D_ARENA_CAP0", "64")
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    ctx.set_pipelining(4, 6)
    for _ in range(2):
        _same(base, ctx.match_resident(0, n), "chunks + arena retries")
    ctx.match_resident(0, n, device_only=True)
    dev = ctx.fetch_device_output()
    for k in ARRAYS:
        assert np.array_equal(np.ascontiguousarray(dev[k]).view(np.uint8), np.ascontiguousarray(base[k]).view(np.uint8)), k
    ctx.close()


def test_new_seeds_reach_every_lane(have_gpu):
    """eg3d_upload_seeds replaces the resident seeds of the context; the lanes created by an earlier call must see the
    new ones (they share the owner's seed buffers per call, not per creation)."""
    s1 = host.Synth(1)
    off, view, xy = s1.seeds_np()
    # other seeds on the same scene: the same tracks in reverse order
    n = len(off) - 1
    order = np.arange(n)[::-1]
    lens = np.diff(off)[order]
    off2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    idx = np.concatenate([np.arange(off[i], off[i + 1]) for i in order]) if n else np.zeros(0, np.int64)
    s2 = host.SeedsArrays(off2, view[idx], xy[idx])
    ctx = api.Context(s1.scene)
    ctx.set_pipelining(4, 4)
    a = ctx.match_refpoints(s1.seeds)
    b = ctx.match_refpoints(C.byref(s2.c), 0, n)
    a2 = ctx.match_refpoints(s1.seeds)
    _same(a, a2, "first seeds again")
    ctx.set_pipelining(1, 0)
    _same(b, ctx.match_refpoints(C.byref(s2.c), 0, n), "second seeds, uncut")
    assert b["n_points"] == a["n_points"] and not np.array_equal(a["key"], b["key"])
    ctx.close()


@pytest.mark.parametrize("unit", [1, 3, 8])
def test_failing_unit_fails_the_call_and_the_context_stays_usable(have_gpu, monkeypatch, unit):
    """A unit that fails when its turn to place comes (EG3D_TEST_FAIL_UNIT, a test knob: the first, a middle or the last
    one of 8) makes the whole call return its error — no lane is left waiting for a turn that never comes — and the next
    call on the same context is complete again."""
    s = host.Synth(1)
    monkeypatch.setenv("EG3D_TEST_FAIL_UNIT", str(unit))
    ctx = api.Context(s.scene)
    monkeypatch.delenv("EG3D_TEST_FAIL_UNIT")   # (read once, at eg3d_create; one failure per context)
    ctx.upload_seeds(s.seeds)
    ctx.set_pipelining(4, 8)
    with pytest.raises(api.Eg3dError) as ei:
        ctx.match_resident(0, s.n_seeds)
    assert "EG3D_TEST_FAIL_UNIT" in str(ei.value)
    with pytest.raises(api.Eg3dError):
        ctx.match_resident(0, s.n_seeds, device_only=True)
    ctx.close()
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    ctx.set_pipelining(4, 8)
    r = ctx.match_resident(0, s.n_seeds)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    assert compare_edgepoints(ref, r)["ok"]
    ctx.close()
