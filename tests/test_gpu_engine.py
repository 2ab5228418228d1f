"""-m gpu: the lane-per-chain engine form of the expand stage (k3c_engine, eg3d_k3c_engine.h). Round 5 built it as
the structural alternative to one wavefront per chain; it is bit-exact and 2.4x slower, so since round 6
the PRODUCT libraries are built without it and these tests run against the variant that carries it
(edgegraph3d_amd/variants/libeg3d_engine.so = the default 6x4 form + -DEG3D_WITH_K3C_ENGINE, built by
__graft_entry__.build()). They keep a second complete implementation of rows a10-a16 — and the state machine of
eg3d_chain_sm.h it drives, which tests/hostsim checks on the CPU — honest. One form, one library: not multiplied by the
DLT forms like the rest of the GPU suite."""
import os

import numpy as np
import pytest

from edgegraph3d_amd import api, build, host
from parity_util import compare_edgepoints

pytestmark = pytest.mark.gpu


def _oracle(scene):
    from oracle import binding as ob
    return ob.Oracle(scene)


@pytest.fixture(autouse=True)
def engine_library(eg3d_form, monkeypatch):
    """Every test here loads the engine variant (form 6x4); the second parametrisation of the suite is skipped."""
    if eg3d_form != 3:
        pytest.skip("the engine variant is built in the default DLT form only")
    assert os.path.exists(build.HIP_LIB_ENGINE), "variants/libeg3d_engine.so is not built (python -m edgegraph3d_amd.build)"
    monkeypatch.setenv("EG3D_LIB", build.HIP_LIB_ENGINE)
    old = api._LIB
    api._LIB = None
    yield
    api._LIB = old


@pytest.fixture(scope="module")
def have_gpu():
    assert api.device_count() >= 1, "no HIP device: the product path has no CPU fallback"


def test_product_library_refuses_the_engine(have_gpu, monkeypatch):
    """EG3D_K3B_ENGINE=1 on a library built without the engine is an error at eg3d_create, not a silent k3b_expand run."""
    monkeypatch.setenv("EG3D_LIB", build.HIP_LIB)
    api._LIB = None
    monkeypatch.setenv("EG3D_K3B_ENGINE", "1")
    s = host.Synth(0)
    with pytest.raises(api.Eg3dError) as ei:
        api.Context(s.scene)
    assert "EG3D_WITH_K3C_ENGINE" in str(ei.value)
    api._LIB = None


@pytest.mark.parametrize("lanes", [0, 3])
def test_lane_per_chain_engine_form_of_the_expand_stage_matches_oracle(have_gpu, monkeypatch, lanes):
    """EG3D_K3B_ENGINE=1 runs the expand stage as the lane-per-chain engine (k3c_engine, eg3d_k3c_engine.h: one lane owns
    a chain — the state machine of eg3d_chain_sm.h — and the wave serves all chains' Gauss-Newton solves and candidate
    searches densely) instead of one wavefront per chain. Measured slower in round 5 and therefore not the default
    (DESIGN_LOG.md), but it is a complete second implementation of rows a10-a16 and must stay bit-exact: C2-sized and
    small scenes, a fuzz scene with mutated polylines, both with the default number of owning lanes per wave and with 3
    (chains queue up behind each other on a lane)."""
    monkeypatch.setenv("EG3D_K3B_ENGINE", "1")
    if lanes:
        monkeypatch.setenv("EG3D_K3C_LANES", str(lanes))
    for cfg in (1, 2):
        s = host.Synth(cfg)
        ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
        ctx = api.Context(s.scene)
        got = ctx.match_refpoints(s.seeds)
        ctx.close()
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (cfg, rep["msgs"][:3])
        assert got["flags"] == ref["flags"] and got["times"]["bytes_algorithmic"] == ref["stats"]["bytes_algorithmic"]
        if cfg == 1:  # the pipelines 1-2 extractor feeds the same stage (a sample = a virtual seed over all views)
            n_sets, row_off, ids = s.polyline_sets(3)
            ctx = api.Context(s.scene)
            gs = ctx.match_polyline_sets(n_sets, row_off, ids)
            ctx.close()
            rs = _oracle(s.scene).match_polyline_sets(n_sets, row_off, ids, nthreads=8)
            rep = compare_edgepoints(rs, gs)
            assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], ("sets", rep["msgs"][:3])
    import ctypes as C
    from fuzz_scenes import draw
    for case in (3, 20):
        _, sa, seeds = draw(case)
        n = len(seeds.trk_off) - 1
        from oracle import binding as ob
        ref = ob.Oracle(C.byref(sa.c)).match(C.byref(seeds.c), 0, n, 8)
        ctx = api.Context(C.byref(sa.c))
        got = ctx.match_refpoints(C.byref(seeds.c), 0, n)
        ctx.close()
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (case, rep["msgs"][:3])



def test_engine_redoes_a_solve_longer_than_a_packed_round_with_its_long_build(have_gpu, monkeypatch):
    """The engine without the solver's long-request path raises CTR_LONG_REFUSED on a solve of more than 32 rows; the host
    latches the general build and redoes the chunk (as for k3b_expand: tests/test_gpu_parity.py)."""
    cfg = host.default_config(1)
    cfg.n_views, cfg.n_seeds, cfg.n_curves, cfg.max_track = 40, 60, 14, 12
    s = host.Synth(cfg)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    assert int(np.diff(ref["obs_off"].astype(np.int64)).max()) > 33
    monkeypatch.setenv("EG3D_K3B_ASSUME_SHORT", "1")
    monkeypatch.setenv("EG3D_K3B_ENGINE", "1")
    ctx = api.Context(s.scene)
    for _ in range(2):
        got = ctx.match_refpoints(s.seeds)
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"][:3]
        assert got["times"]["bytes_algorithmic"] == ref["stats"]["bytes_algorithmic"]
    ctx.close()
