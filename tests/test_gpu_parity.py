"""-m gpu: the parity tests proper. The HIP path is called through the C ABI and compared with
the CPU oracle on the same seeded inputs (exact structure; X within 1e-4 relative as
BASELINE.json's north_star states — in practice the arithmetic contract makes it bit-exact)."""
import os

import numpy as np
import pytest

from edgegraph3d_amd import api, host
from parity_util import compare_edgepoints

pytestmark = pytest.mark.gpu


def _oracle(scene):
    from oracle import binding as ob
    return ob.Oracle(scene)


@pytest.fixture(scope="module")
def have_gpu():
    assert api.device_count() >= 1, "no HIP device: the product path has no CPU fallback"


@pytest.mark.parametrize("cfg", [0, 1])
def test_grids_match_oracle(have_gpu, cfg):
    s = host.Synth(cfg)
    ctx = api.Context(s.scene)
    o = _oracle(s.scene)
    for v in range(s.n_views):
        for which in (0, 1):
            a, b = ctx.grid(v, which), o.grid(v, which)
            assert a[0] == b[0] and a[1] == b[1]
            assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    ctx.close()


@pytest.mark.parametrize("cfg", [0, 1])
def test_stage_a_candidates_exact(have_gpu, cfg):
    s = host.Synth(cfg)
    ctx = api.Context(s.scene)
    got = ctx.candidates(s.seeds, 0, s.n_seeds)
    ref = _oracle(s.scene).candidates(s.seeds, 0, s.n_seeds)
    for k in ("n_sv", "n_tasks"):
        assert got[k] == ref[k], k
    for k in ("cand_off", "cand_pl", "start_off", "start_pl", "start_seg", "task_sv", "task_hit", "task_list_off",
              "list_off", "hit_pl", "hit_seg"):
        assert np.array_equal(got[k], ref[k]), k
    for k in ("start_xy", "hit_xy"):
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k + " not bit-exact"
    ctx.close()


@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_full_path_parity(have_gpu, cfg):
    s = host.Synth(cfg)
    ctx = api.Context(s.scene)
    got = ctx.match_refpoints(s.seeds)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"], rep["msgs"]
    assert got["n_chains"] == ref["stats"]["n_chains"]
    assert got["n_tasks"] == ref["stats"]["n_tasks"]
    assert (got["flags"] & 7) == 0, "device capacity flag raised"
    ctx.close()


@pytest.mark.parametrize("n_views", [0, 31, 70], ids=["small scene", "31 views", "70 views"])
def test_scene_class_builds_of_the_expand_kernel_agree_with_the_general_one(have_gpu, monkeypatch, n_views):
    """libeg3d.so carries three instantiations of k3b_expand, chosen per context by the scene's class (launch_k3b): small
    scenes (<= 28 views, polylines of <= 512 vertices: no long-request solver path, no unstaged side walks), many views
    (>= 29: chain following one step at a time, windowed central pre-solves), and the general one. EG3D_K3B_FULL=1
    (read once, by eg3d_create) forces the general kernel: both must give the same cloud bit for bit — on the seed path
    and on the polyline-set path — and the small build must not have met a request it cannot solve (the call would fail)."""
    if n_views:
        cfg = host.default_config(1)
        cfg.rng_seed, cfg.n_views, cfg.n_curves, cfg.max_track, cfg.n_seeds = 13 + n_views, n_views, 24, n_views, 80
        s = host.Synth(cfg)
    else:
        s = host.Synth(1)
    small = api.Context(s.scene)
    a = small.match_refpoints(s.seeds)
    n_sets, row_off, ids = s.polyline_sets(3)
    sa = small.match_polyline_sets(n_sets, row_off, ids)
    small.close()
    monkeypatch.setenv("EG3D_K3B_FULL", "1")
    general = api.Context(s.scene)
    b = general.match_refpoints(s.seeds)
    sb = general.match_polyline_sets(n_sets, row_off, ids)
    general.close()
    for x, y in ((a, b), (sa, sb)):
        assert x["n_points"] == y["n_points"] > 0 and x["n_obs"] == y["n_obs"]
        for k in ("obs_off", "key", "obs_view", "obs_pl", "obs_seg"):
            assert np.array_equal(x[k], y[k]), k
        assert np.array_equal(x["X"].view(np.uint32), y["X"].view(np.uint32))
        assert np.array_equal(x["obs_xy"].view(np.uint32), y["obs_xy"].view(np.uint32))
        assert x["flags"] == y["flags"]


def test_solve_longer_than_a_packed_round_is_redone_by_the_general_build(have_gpu, monkeypatch):
    """A few-views build of the expand stage (k3b_expand small scenes / the engine without the solver's long-request
    path) that meets a solve of more than 32 rows must not fail the call: it raises CTR_LONG_REFUSED, the host latches
    the general build for the context and redoes the chunk (eg3d_api.hip run_stage_b). Forced here with
    EG3D_K3B_ASSUME_SHORT=1 on a 40-view scene whose points carry up to 40 observations; cloud == oracle, twice (the
    second call runs the latched general build from the start)."""
    cfg = host.default_config(1)
    cfg.n_views, cfg.n_seeds, cfg.n_curves, cfg.max_track = 40, 60, 14, 12
    s = host.Synth(cfg)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    assert int(np.diff(ref["obs_off"].astype(np.int64)).max()) > 33
    monkeypatch.setenv("EG3D_K3B_ASSUME_SHORT", "1")
    ctx = api.Context(s.scene)
    for _ in range(2):
        got = ctx.match_refpoints(s.seeds)
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"][:3]
        assert got["times"]["bytes_algorithmic"] == ref["stats"]["bytes_algorithmic"]
    ctx.close()


def test_seed_range_concatenation(have_gpu):
    """Ranges are independent: [0,n/2) + [n/2,n) == [0,n) (the multi-GPU sharding property)."""
    s = host.Synth(1)
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    n = s.n_seeds
    full = ctx.match_resident(0, n)
    a = ctx.match_resident(0, n // 2)
    b = ctx.match_resident(n // 2, n)
    assert a["n_points"] + b["n_points"] == full["n_points"]
    assert np.array_equal(np.concatenate([a["X"], b["X"]]).view(np.uint32), full["X"].view(np.uint32))
    assert np.array_equal(np.concatenate([a["key"], b["key"]]), full["key"])
    ctx.close()


def test_gn_filter_parity(have_gpu):
    s = host.Synth(1)
    X, off, view, xy = s.points(20000)
    ctx = api.Context(s.scene)
    Xo, inl, ms = ctx.gn_filter(X, off, view, xy, 3.0)
    Xr, ir = _oracle(s.scene).gn_filter(X, off, view, xy, 3.0, nthreads=8)
    assert np.array_equal(inl, ir)
    assert np.array_equal(Xo.view(np.uint32), Xr.view(np.uint32))
    assert 0.5 < inl.mean() < 1.0
    ctx.close()
    # the 16-view C5 rig (k really spans 3..10), both abs() behaviours of Q9, and lists long enough that
    # a block's observation slice does NOT fit its LDS staging area (the HBM-operand fallback)
    s = host.Synth(5)
    X, off, view, xy = s.points(60000)
    assert (np.diff(off) == 10).any() and (np.diff(off) == 3).any()
    ctx = api.Context(s.scene)
    for legacy in (False, True):
        Xo, inl, ms = ctx.gn_filter(X, off, view, xy, 2.25, legacy_abs=legacy)
        Xr, ir = _oracle(s.scene).gn_filter(X, off, view, xy, 2.25, legacy_abs=legacy, nthreads=8)
        assert np.array_equal(inl, ir) and np.array_equal(Xo.view(np.uint32), Xr.view(np.uint32))
    # 300 points x 16 observations (every view twice over would exceed the rig: reuse views) > 2816 per block
    n = 600
    big_off = np.arange(n + 1, dtype=np.uint32) * 16
    big_view = np.tile(np.arange(16, dtype=np.int32), n)
    P = s.scene_np()["cam_P"].reshape(16, 4, 4)[:, :3, :].astype(np.float64)
    rng = np.random.default_rng(3)
    Xt = rng.uniform(-100, 100, (n, 3))
    h = np.einsum("vij,nj->nvi", P, np.concatenate([Xt, np.ones((n, 1))], 1))
    big_xy = (h[:, :, :2] / h[:, :, 2:3] + rng.normal(0, 0.5, (n, 16, 2))).astype(np.float32).reshape(-1, 2)
    Xs = (Xt + rng.normal(0, 2.0, Xt.shape)).astype(np.float32)
    Xo, inl, ms = ctx.gn_filter(Xs, big_off, big_view, big_xy, 2.25)
    Xr, ir = _oracle(s.scene).gn_filter(Xs, big_off, big_view, big_xy, 2.25, nthreads=8)
    assert np.array_equal(inl, ir) and np.array_equal(Xo.view(np.uint32), Xr.view(np.uint32)) and inl.mean() > 0.5
    ctx.close()


def test_gn_filter_full_config5_one_million_points(have_gpu):
    """BASELINE configs[4] at its FULL size: 1 000 000 edge-points of the 16-view rig (k in 3..10, 5.76 M
    observations), both abs() behaviours of Q9 — X and the inlier flags of every point bit-exact."""
    s = host.Synth(5)
    X, off, view, xy = s.points(1000000)
    assert len(off) - 1 == 1000000
    ctx = api.Context(s.scene)
    o = _oracle(s.scene)
    for legacy in (False, True):
        Xo, inl, ms = ctx.gn_filter(X, off, view, xy, 2.25, legacy_abs=legacy)
        Xr, ir = o.gn_filter(X, off, view, xy, 2.25, legacy_abs=legacy, nthreads=os.cpu_count())
        assert np.array_equal(inl, ir) and np.array_equal(Xo.view(np.uint32), Xr.view(np.uint32))
        assert 0.3 < inl.mean() < 1.0
    ctx.close()


@pytest.mark.timeout(1200)
def test_c4_far_end_window_parity(have_gpu):
    """BASELINE configs[3], the LAST 1200 of its 100 000 seeds (98 800 .. 100 000) on the full scene and the
    full resident seed set: global seed indices near the end of the range, a window the first-seeds test never
    touches. Bit-exact, ids and order included."""
    s = host.Synth(4)
    assert s.n_seeds == 100000
    lo, hi = s.n_seeds - 1200, s.n_seeds
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    got = ctx.match_resident(lo, hi)
    ref = _oracle(s.scene).match(s.seeds, lo, hi, nthreads=os.cpu_count())
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"]
    assert (got["flags"] & 7) == 0 and got["n_points"] > 300000
    assert got["key"][:, 0].min() >= lo and got["key"][:, 0].max() < hi
    ctx.close()


def test_c4_shaped_subset_parity_and_capacity_growth(have_gpu, monkeypatch, capfd):
    """200 views / ~20k segments per view (BASELINE configs[3] shape), first 40 seeds: points carry
    dozens of observations, which outgrows the default per-chain pool -> the library must enlarge
    it and still match the oracle exactly. Since round 6 only the chains that overflowed are launched again
    (the others keep their packed results): the trace says so, and the byte counter — of which the relaunched
    chains' first attempt is taken back — still equals the oracle's."""
    cfg = host.default_config(4)
    cfg.n_seeds = 40
    s = host.Synth(cfg)
    monkeypatch.setenv("EG3D_TRACE_ARENA", "1")
    ctx = api.Context(s.scene)
    got = ctx.match_refpoints(s.seeds)
    err = capfd.readouterr().err
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=16)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"], rep["msgs"]
    assert (got["flags"] & 7) == 0 and got["flags"] == ref["flags"]
    assert got["n_obs"] > 20 * got["n_points"]
    assert "outgrew their slices" in err and "relaunching those" in err, err[-400:]
    assert got["times"]["bytes_algorithmic"] == ref["stats"]["bytes_algorithmic"]
    again = ctx.match_refpoints(s.seeds)   # the grown capacities are kept: no relaunch the second time
    assert "outgrew" not in capfd.readouterr().err
    assert np.array_equal(again["X"].view(np.uint32), got["X"].view(np.uint32))
    ctx.close()


@pytest.mark.parametrize("lanes", [1, 3])
def test_relaunch_of_overflowed_chains_round_after_round(have_gpu, monkeypatch, capfd, lanes):
    """Tiny initial capacities (EG3D_CHAIN_CAP0 / EG3D_POOL_CAP0, test knobs) make most chains of a C2-sized scene outgrow
    their slices several times over: every round relaunches only what overflowed in the round before, with doubled
    capacities, until nothing does. Cloud, flags and byte counter == oracle; also with the call cut into units on lanes."""
    s = host.Synth(1)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    monkeypatch.setenv("EG3D_CHAIN_CAP0", "8")
    monkeypatch.setenv("EG3D_POOL_CAP0", "64")
    monkeypatch.setenv("EG3D_TRACE_ARENA", "1")
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    ctx.set_pipelining(lanes, 0)
    got = ctx.match_resident(0, s.n_seeds)
    err = capfd.readouterr().err
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"][:3]
    assert got["flags"] == ref["flags"] and got["times"]["bytes_algorithmic"] == ref["stats"]["bytes_algorithmic"]
    assert err.count("relaunching those") >= 3, err[-600:]
    dev = ctx.match_resident(0, s.n_seeds, device_only=True)
    assert dev["n_points"] == ref["n_points"]
    ctx.close()


@pytest.mark.timeout(1200)
def test_c4_1200_seeds_parity(have_gpu):
    """BASELINE configs[3] shape at a size the oracle finishes in about a minute on the box's cores: the
    first 1200 seeds of the 200-view / 20k-segments-per-view scene (~0.7 M edge-points carrying ~70
    observations each: the regime of the lane-group solver's chunked long lists). Bit-exact, ids and
    order included; plus the properties that hold at any size (chains are runs of one key, obs_off is a
    prefix sum, every observation's view is unique within its point)."""
    cfg = host.default_config(4)
    cfg.n_seeds = 1200
    s = host.Synth(cfg)
    ctx = api.Context(s.scene)
    got = ctx.match_refpoints(s.seeds)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=os.cpu_count())
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"]
    assert (got["flags"] & 7) == 0 and got["n_points"] > 300000
    off = got["obs_off"].astype(np.int64)
    assert (np.diff(off) >= 3).all() and off[-1] == got["n_obs"]
    k = got["key"]
    new_chain = np.any(k[1:, :3] != k[:-1, :3], axis=1)
    assert ((k[1:, 3] == k[:-1, 3] + 1) | new_chain).all() and (k[1:, 3][new_chain] == 0).all()
    # views are unique within a point (checked on a sample of points)
    for i in np.random.default_rng(0).integers(0, got["n_points"], 2000):
        v = got["obs_view"][off[i]:off[i + 1]]
        assert len(np.unique(v)) == len(v)
    ctx.close()


def test_contexts_in_concurrent_threads(have_gpu):
    """bench.py keeps several steps in flight: one context per host thread on the same device.
    Concurrent contexts must give exactly what one context gives alone."""
    import threading
    s = host.Synth(1)
    n = s.n_seeds
    ref_ctx = api.Context(s.scene)
    ref_ctx.upload_seeds(s.seeds)
    ranges = [(0, n), (0, n // 2), (n // 2, n)]
    ref = [ref_ctx.match_resident(b, e) for b, e in ranges]
    out = [None] * len(ranges)
    errs = []
    clones = [ref_ctx.clone(), ref_ctx.clone()]  # eg3d_clone: share the resident scene and seeds
    ref_ctx.close()                               # the clones keep the shared buffers alive

    def work(i):
        try:
            if i < len(clones):
                ctx = clones[i]
            else:
                ctx = api.Context(s.scene)
                ctx.upload_seeds(s.seeds)
            for _ in range(3):
                out[i] = ctx.match_resident(*ranges[i])
            ctx.close()
        except Exception as ex:  # noqa: BLE001
            errs.append(ex)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(ranges))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for r, g in zip(ref, out):
        assert g["n_points"] == r["n_points"] and g["n_obs"] == r["n_obs"]
        assert np.array_equal(g["X"].view(np.uint32), r["X"].view(np.uint32))
        assert np.array_equal(g["key"], r["key"])
        assert np.array_equal(g["obs_xy"].view(np.uint32), r["obs_xy"].view(np.uint32))


def test_reference_surface_shim_on_gpu(have_gpu, tmp_path):
    """include/eg3d_refapi.hpp driven the way reference-side C++ host code would: SfMData /
    PolyLineGraph2D / F in, plg_matching_from_refpoints_parallel out — with three batches in flight
    on clones — equals one direct C-ABI call bit for bit (tests/refapi/refapi_check.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "edgegraph3d_amd")
    exe = str(tmp_path / "refapi_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "refapi", "refapi_check.cpp"), "-L", pkg, "-leg3d", "-leg3d_host",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-lamdhip64", "-o", exe])
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "\nOK points=" in "\n" + out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_polyline_sets_extractor_parity(have_gpu, cfg):
    """SURVEY N1 (pipelines 1-2): polyline sets -> 20 px samples -> epipolar hits on the set ->
    3-view consensus + expand-all-views, against the oracle's restatement of
    polyline_matching.cpp:45-73,153-208. Sets = the polylines of each synthetic 3-D curve."""
    s = host.Synth(cfg)
    n, row_off, ids = s.polyline_sets()
    ctx = api.Context(s.scene)
    got = ctx.match_polyline_sets(n, row_off, ids)
    ref = _oracle(s.scene).match_polyline_sets(n, row_off, ids, nthreads=16)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"], rep["msgs"]
    assert got["n_tasks"] == ref["stats"]["n_tasks"] and got["n_chains"] == ref["stats"]["n_chains"]
    assert got["n_points"] > 0 and (got["flags"] & 7) == 0
    assert got["times"]["bytes_algorithmic"] == ref["stats"]["bytes_algorithmic"]
    # a sub-range of sets is the corresponding slice of the whole (sets are independent)
    if n >= 4:
        part = ctx.match_polyline_sets(n, row_off, ids, 1, 3)
        refp = _oracle(s.scene).match_polyline_sets(n, row_off, ids, 1, 3, nthreads=8)
        assert compare_edgepoints(refp, part, rel_tol=1e-4)["ok"]
    ctx.close()


def test_full_size_properties_dtu006_shaped(have_gpu):
    """BASELINE-size run (C3': 25 views / 6268 seeds / ~15k segments per view, 1.7 M edge-points) checked
    through size-independent properties instead of the oracle: run-to-run determinism (bitwise),
    seed-range concatenation, ordering of the emission keys, well-formed observation lists, and a
    40-seed window of it against the oracle."""
    s = host.Synth(3)
    n = s.n_seeds
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    full = ctx.match_resident(0, n)
    again = ctx.match_resident(0, n)
    for k in ("X", "obs_xy"):
        assert np.array_equal(full[k].view(np.uint32), again[k].view(np.uint32)), k
    for k in ("key", "obs_off", "obs_view", "obs_pl", "obs_seg"):
        assert np.array_equal(full[k], again[k]), k
    cut = n // 3
    a, b = ctx.match_resident(0, cut), ctx.match_resident(cut, n)
    assert a["n_points"] + b["n_points"] == full["n_points"] > 1000000
    assert np.array_equal(np.concatenate([a["X"], b["X"]]).view(np.uint32), full["X"].view(np.uint32))
    assert np.array_equal(np.concatenate([a["key"], b["key"]]), full["key"])
    key = full["key"].astype(np.int64)
    order = key[:, 0] * (1 << 40) + key[:, 1] * (1 << 30) + key[:, 2] * (1 << 20) + key[:, 3]
    assert (np.diff(order) > 0).all(), "emission order (seed, entry, hit, index) is strictly increasing"
    off = full["obs_off"].astype(np.int64)
    m = np.diff(off)
    assert (m >= 3).all()   # (a view may legitimately appear twice on a point: the oracle agrees on those)
    assert np.isfinite(full["X"]).all()
    assert ((full["obs_view"] >= 0) & (full["obs_view"] < s.n_views)).all()
    assert (full["flags"] & 7) == 0
    lo = 3000
    ref = _oracle(s.scene).match(s.seeds, lo, lo + 40, nthreads=16)
    win = ctx.match_resident(lo, lo + 40)
    assert compare_edgepoints(ref, win, rel_tol=1e-4)["ok"]
    ctx.close()


def test_rccl_allgather_c_abi_single_rank(have_gpu):
    """include/eg3d_rccl.h: the C-ABI all-gather of the cloud (counts, one padded ncclAllGather of the
    packed SoA from the context's HBM buffers, device compaction). With one rank the gathered cloud
    must equal the rank's own output (tests/rccl_single_rank_check.py, in its own process); the
    N-rank ordering logic is covered on CPU by tests/test_multirank_gloo.py."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "rccl_single_rank_check.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "RCCL-GATHER-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_quirks_q4_q12_q13_on_the_device():
    """The HIP path follows the reference at the three stale-state / early-return quirks: it equals
    the oracle's reference behaviour bit for bit on scenes where a "corrected" Q4 / Q12 / Q13
    (oracle test hook orc_set_quirk_fixes) gives a different cloud (tests/test_quirks.py)."""
    from oracle import binding as ob
    L = ob.lib()
    for cfg, lo, hi, bits in ((0, 0, None, (4, 13)), (2, 80, 90, (12,))):
        s = host.Synth(cfg)
        hi = s.n_seeds if hi is None else hi
        ctx = api.Context(s.scene)
        ctx.upload_seeds(s.seeds)
        got = ctx.match_resident(lo, hi)
        o = ob.Oracle(s.scene)
        ref = o.match(s.seeds, lo, hi, 4)
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"], rep
        for q in bits:
            try:
                L.orc_set_quirk_fixes(1 << q)
                fixed = o.match(s.seeds, lo, hi, 4)
            finally:
                L.orc_set_quirk_fixes(0)
            same = (fixed["n_points"] == got["n_points"] and fixed["n_obs"] == got["n_obs"] and
                    np.array_equal(fixed["X"].view(np.uint32), got["X"].view(np.uint32)))
            assert not same, "Q%d: the device output equals the corrected variant" % q
        ctx.close()


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_allgather_c_abi_two_ranks():
    """include/eg3d_rccl.h with TWO ranks (unequal, empty and incomplete shards): needs >= 2 GPUs, so it
    is skipped on the single-GPU boxes; tests/rccl_two_rank_check.py is the script the driver can run
    on a multi-GPU node."""
    import subprocess
    import sys
    if api.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_two_rank_check.py")],
                       capture_output=True, text=True)
    assert p.returncode == 0 and "RCCL-2RANK-OK" in p.stdout, p.stdout + p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_scenes_built_from_edge_images_parity():
    """SURVEY N2 feeding the path: (1) 25 real dtu006 edge maps -> polyline graphs, with synthetic look-at
    cameras at the listed centres (real polyline statistics; geometry not consistent with the images),
    (2) a synthetic scene rasterised into edge images and rebuilt by the same builder (consistent
    geometry). HIP path == oracle bit for bit on both."""
    import ctypes as C
    import real_scene as rs
    from oracle import binding as ob
    for name, (sc, seeds, info) in (("real", rs.real_edges_scene(n_seeds=3000)), ("rendered", rs.rendered_edges_scene(1))):
        ctx = api.Context(C.byref(sc.c))
        got = ctx.match_refpoints(C.byref(seeds.c), 0, int(seeds.c.n_seeds))
        ref = ob.Oracle(C.byref(sc.c)).match(C.byref(seeds.c), 0, int(seeds.c.n_seeds), os.cpu_count())
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (name, rep["msgs"])
        assert got["n_points"] > 1000, name
        assert (got["flags"] & 7) == 0
        ctx.close()


def test_device_grids_equal_the_host_builder(have_gpu, monkeypatch):
    """K0 (round 6): eg3d_create builds both uniform grids on the device — one lane walks one polyline and emits
    (view, cell, polyline) keys; sort + unique + one CSR pass. The host builder (eg3d_host_build_grid: the same walk, one
    view at a time; what rounds 1-5 shipped) and the oracle are the statement it must reproduce exactly: the C2 scene, a
    quarter of C3', and mutated fuzz scenes (loops, invalidated polylines that keep their vertices, small images), every
    view, both cell sizes; EG3D_GRID_ON_HOST=1 (the diagnostic host path) gives the same context."""
    import ctypes as C
    from fuzz_scenes import draw
    scenes = [(host.Synth(2).scene, "C2", None)]
    s3 = host.Synth(3)
    scenes.append((s3.scene, "C3'", range(0, s3.n_views, 4)))
    keep = [s3]
    for case in (1, 3, 6, 9, 20, 33):
        _, sa, _ = draw(case)
        keep.append(sa)
        scenes.append((C.byref(sa.c), "fuzz %d" % case, None))
    for scene, name, views in scenes:
        ctx = api.Context(scene)
        sc = scene.contents if hasattr(scene, "contents") else scene._obj
        for v in (views if views is not None else range(int(sc.n_views))):
            for which, cell in ((0, 30.0), (1, 4.0)):
                a = ctx.grid(v, which)
                ncols, nrows, off, ids, _ = host.build_grid(scene, v, cell)
                assert a[0] == ncols and a[1] == nrows, (name, v, which)
                assert np.array_equal(a[2], off) and np.array_equal(a[3], ids), (name, v, which)
        ctx.close()
    monkeypatch.setenv("EG3D_GRID_ON_HOST", "1")
    s = host.Synth(1)
    ch = api.Context(s.scene)
    monkeypatch.delenv("EG3D_GRID_ON_HOST")
    cd = api.Context(s.scene)
    for v in range(s.n_views):
        for which in (0, 1):
            a, b = ch.grid(v, which), cd.grid(v, which)
            assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    assert compare_edgepoints(ref, ch.match_refpoints(s.seeds))["ok"]
    ch.close()
    cd.close()
