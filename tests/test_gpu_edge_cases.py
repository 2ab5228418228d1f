"""-m gpu: edge cases of the C ABI against the oracle — empty and ragged inputs, tracks that are too
short or repeat a view (Q2), seeds far from every edge, invalid F pairs, invalid arguments."""
import ctypes as C
import os

import numpy as np
import pytest

from edgegraph3d_amd import _cdefs as D
from edgegraph3d_amd import api, host
from parity_util import compare_edgepoints

pytestmark = pytest.mark.gpu


def _oracle(scene):
    from oracle import binding as ob
    return ob.Oracle(scene)


def _both(scene_ptr, seeds_ptr, n):
    ctx = api.Context(scene_ptr)
    got = ctx.match_refpoints(seeds_ptr, 0, n)
    ref = _oracle(scene_ptr).match(seeds_ptr, 0, n, nthreads=4)
    ctx.close()
    return got, ref


def test_empty_ranges_and_zero_seeds():
    s = host.Synth(0)
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    r = ctx.match_resident(5, 5)
    assert r["n_points"] == 0 and r["n_obs"] == 0 and len(r["obs_off"]) == 1
    empty = host.SeedsArrays(np.zeros(1, np.uint32), np.zeros(0, np.int32), np.zeros((0, 2), np.float32))
    r = ctx.match_refpoints(C.byref(empty.c), 0, 0)
    assert r["n_points"] == 0
    n, row_off, ids = s.polyline_sets()
    r = ctx.match_polyline_sets(n, row_off, ids, 2, 2)
    assert r["n_points"] == 0
    # a set with no polylines at all
    r = ctx.match_polyline_sets(1, np.zeros(s.n_views + 1, np.uint32), np.zeros(0, np.uint32))
    assert r["n_points"] == 0 and r["n_tasks"] == 0
    ctx.close()


def test_ragged_tracks_short_duplicate_views_and_far_seeds():
    """Mutated seed set: tracks cut to 1-2 views (no triple possible), a view repeated inside a track
    (the LAST observation of a view wins, Q2), observations moved far from every polyline (no start
    hits), and untouched seeds in between."""
    s = host.Synth(1)
    off, view, xy = s.seeds_np()
    rng = np.random.default_rng(7)
    noff, nview, nxy = [0], [], []
    for i in range(len(off) - 1):
        v = list(view[off[i]:off[i + 1]])
        p = [tuple(q) for q in xy[off[i]:off[i + 1]]]
        kind = i % 5
        if kind == 1:
            v, p = v[:1 + i % 2], p[:1 + i % 2]
        elif kind == 2 and len(v) >= 3:
            v.append(v[0])                       # repeat the first view with a shifted observation
            p.append((p[0][0] + 3.0, p[0][1] - 2.0))
        elif kind == 3:
            p = [(float(rng.uniform(5, 40)), float(rng.uniform(5, 40))) for _ in p]   # image corner: nothing nearby
        nview += v
        nxy += p
        noff.append(len(nview))
    seeds = host.SeedsArrays(np.asarray(noff, np.uint32), np.asarray(nview, np.int32), np.asarray(nxy, np.float32))
    got, ref = _both(s.scene, C.byref(seeds.c), len(noff) - 1)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"], rep["msgs"]
    assert got["n_points"] > 0
    keys = set(int(k) for k in got["key"][:, 0])
    assert not any(k % 5 == 1 for k in keys), "a track of one or two views cannot yield a 3-view hypothesis"


def test_invalid_fundamental_matrices_disable_their_pairs():
    """F_valid = 0 (the reference's 1x1 Mat => computeCorrespondEpilineSinglePoint fails) for every
    pair that involves view 1: the view drops out of all epipolar searches; oracle and GPU agree."""
    s = host.Synth(1)
    sc = s.scene_np()
    Fv = sc["F_valid"].copy()
    Fv[1, :] = 0
    Fv[:, 1] = 0
    sc["F_valid"] = Fv
    sa = host.SceneArrays(sc)
    got, ref = _both(C.byref(sa.c), s.seeds, s.n_seeds)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"], rep["msgs"]
    full = api.Context(s.scene)
    whole = full.match_refpoints(s.seeds, 0, s.n_seeds)
    full.close()
    assert got["n_points"] != whole["n_points"]


def test_invalid_arguments_are_reported_not_crashing():
    s = host.Synth(0)
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    with pytest.raises(api.Eg3dError):
        ctx.match_resident(0, s.n_seeds + 1)                      # range beyond the resident seeds
    off, view, xy = s.seeds_np()
    bad = view.copy()
    bad[0] = s.n_views                                            # view id out of range
    seeds = host.SeedsArrays(off, bad, xy)
    with pytest.raises(api.Eg3dError):
        ctx.upload_seeds(C.byref(seeds.c))
    n, row_off, ids = s.polyline_sets()
    bad_ids = ids.copy()
    bad_ids[0] = 10 ** 6
    with pytest.raises(api.Eg3dError):
        ctx.match_polyline_sets(n, row_off, bad_ids)
    assert b"polyline id out of range" in api.lib().eg3d_last_error()
    # the context is still usable after the errors
    ctx.upload_seeds(s.seeds)
    assert ctx.match_resident(0, s.n_seeds)["n_points"] > 0
    ctx.close()


@pytest.mark.parametrize("rng_seed,n_views,n_curves,max_track", [(11, 5, 9, 5), (12, 12, 20, 12), (13, 31, 24, 31),
                                                               (14, 70, 30, 70)])
def test_random_scenes_parity(rng_seed, n_views, n_curves, max_track):
    """Other scene shapes than the named configs (few views; tracks as long as the rig; more than 64
    views so points outgrow a wavefront's 64 rows and the LDS candidate lists): both extractors
    against the oracle, bit-exact, including with the K3a work queue forced on."""
    cfg = host.default_config(1)
    cfg.rng_seed = rng_seed
    cfg.n_views = n_views
    cfg.n_curves = n_curves
    cfg.max_track = max_track
    cfg.n_seeds = 160 if n_views < 40 else 60
    s = host.Synth(cfg)
    ctx = api.Context(s.scene)
    o = _oracle(s.scene)
    got = ctx.match_refpoints(s.seeds)
    ref = o.match(s.seeds, 0, s.n_seeds, nthreads=16)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"] and rep["bitexact_X"], rep["msgs"]
    assert (got["flags"] & 7) == 0 and got["n_points"] > 0
    n, row_off, ids = s.polyline_sets(4)
    gs = ctx.match_polyline_sets(n, row_off, ids)
    rs = o.match_polyline_sets(n, row_off, ids, nthreads=16)
    rep = compare_edgepoints(rs, gs, rel_tol=1e-4)
    assert rep["ok"] and rep["bitexact_X"], rep["msgs"]
    ctx.close()


def test_very_long_polylines_parity():
    """Polylines of more than 512 vertices (the side-walk LDS staging limit) and walks that test more
    than 64 segments per step: the tiny scene with every segment split into 14 collinear pieces."""
    s = host.Synth(0)
    sc = s.scene_np()
    K = 14
    pvo, vtx = sc["pl_vtx_off"], sc["vtx_xy"]
    new_off, new_vtx = [0], []
    for p in range(len(pvo) - 1):
        v = vtx[pvo[p]:pvo[p + 1]]
        if len(v) >= 2:
            for i in range(len(v) - 1):
                for k in range(K):
                    t = np.float32(k) / np.float32(K)
                    new_vtx.append(v[i] + (v[i + 1] - v[i]) * t)
            new_vtx.append(v[-1])
        new_off.append(len(new_vtx))
    sc["pl_vtx_off"] = np.asarray(new_off, np.uint32)
    sc["vtx_xy"] = np.asarray(new_vtx, np.float32).reshape(-1, 2)
    assert np.diff(sc["pl_vtx_off"]).max() > 512
    sa = host.SceneArrays(sc)
    got, ref = _both(C.byref(sa.c), s.seeds, s.n_seeds)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"] and rep["bitexact_X"], rep["msgs"]
    assert got["n_points"] > 100 and (got["flags"] & 7) == 0
    n, row_off, ids = s.polyline_sets(2)
    ctx = api.Context(C.byref(sa.c))
    gs = ctx.match_polyline_sets(n, row_off, ids)
    rs = _oracle(C.byref(sa.c)).match_polyline_sets(n, row_off, ids, nthreads=8)
    assert compare_edgepoints(rs, gs, rel_tol=1e-4)["ok"]
    ctx.close()


def test_crowded_neighbourhoods_more_than_64_candidate_polylines():
    """K1 takes the candidate polylines of a (seed, entry) in batches of 64 ids and scans a batch's segments as one flat
    sequence; K2 deals the segments of all candidates of a list to the lanes the same way. The tiny scene with clutter:
    around the observations of the first seeds every view gets 70-200 extra 2- and 3-vertex polylines within the 30 px
    window (some within the 10 px start radius, some of them degenerate: repeated vertices), so that entries have more
    than 64 and more than 128 candidates. Stage A and the whole path must equal the oracle bit for bit."""
    s = host.Synth(0)
    sc = s.scene_np()
    off, view, xy = s.seeds_np()
    rng = np.random.default_rng(64128)
    V = int(sc["n_views"])
    vpo, pvo, vtx = sc["view_pl_off"], sc["pl_vtx_off"], sc["vtx_xy"]
    node0 = int(max(sc["pl_start"].max(), sc["pl_end"].max())) + 1
    n_vpo, n_pvo, n_vtx, n_st, n_en, n_val = [0], [0], [], [], [], []
    for v in range(V):
        for g in range(int(vpo[v]), int(vpo[v + 1])):   # the view's own polylines, unchanged
            n_vtx.extend(vtx[pvo[g]:pvo[g + 1]])
            n_pvo.append(len(n_vtx))
            n_st.append(sc["pl_start"][g]); n_en.append(sc["pl_end"][g]); n_val.append(sc["pl_valid"][g])
        centres = [xy[e] for p in range(min(6, len(off) - 1)) for e in range(int(off[p]), int(off[p + 1])) if view[e] == v]
        for ci, c in enumerate(centres):
            for k in range(70 if ci % 2 else 200):
                r = np.float32(rng.uniform(0.5, 9.0) if k % 3 == 0 else rng.uniform(9.0, 27.0))
                a = rng.uniform(0, 2 * np.pi)
                p0 = (c + r * np.array([np.cos(a), np.sin(a)])).astype(np.float32)
                d = rng.normal(0, 1.5, 2).astype(np.float32)
                pts = [p0, p0 + d] if k % 5 else [p0, p0, p0 + d]   # every fifth starts with a zero-length segment
                if k % 11 == 0:
                    pts.append(p0 + d + rng.normal(0, 1.5, 2).astype(np.float32))
                n_vtx.extend(pts)
                n_pvo.append(len(n_vtx))
                n_st.append(node0); n_en.append(node0 + 1); n_val.append(1)
                node0 += 2
        n_vpo.append(len(n_pvo) - 1)
    sc["view_pl_off"] = np.asarray(n_vpo, np.uint32)
    sc["pl_vtx_off"] = np.asarray(n_pvo, np.uint32)
    sc["vtx_xy"] = np.asarray(n_vtx, np.float32).reshape(-1, 2)
    sc["pl_start"], sc["pl_end"] = np.asarray(n_st, np.uint32), np.asarray(n_en, np.uint32)
    sc["pl_valid"] = np.asarray(n_val, np.uint8)
    sa = host.SceneArrays(sc)
    ctx, orc = api.Context(C.byref(sa.c)), _oracle(C.byref(sa.c))
    ca, cb = ctx.candidates(s.seeds, 0, s.n_seeds), orc.candidates(s.seeds, 0, s.n_seeds)
    per_entry = np.diff(cb["cand_off"])
    assert per_entry.max() > 128 and (per_entry > 64).sum() >= 4, per_entry.max()
    for k in cb:
        x, y = ca[k], cb[k]
        if isinstance(y, np.ndarray):
            bits = (lambda a: a.view(np.uint32) if a.dtype == np.float32 else a)
            assert np.array_equal(bits(x), bits(y)), k
        else:
            assert x == y, (k, x, y)
    got = ctx.match_refpoints(s.seeds, 0, s.n_seeds)
    ref = orc.match(s.seeds, 0, s.n_seeds, nthreads=8)
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], rep["msgs"][:3]
    assert got["flags"] == ref["flags"] and got["times"]["bytes_algorithmic"] == ref["stats"]["bytes_algorithmic"]
    ctx.close()


def test_hypothesis_arena_overflow_is_retried(monkeypatch):
    """The bump-allocated arena of hypothesis point lists starts from an estimate; when a batch
    outgrows it the kernels flag the overflow and the stage is rerun with a larger arena. Forced
    here with a tiny initial arena (EG3D_ARENA_CAP0): both kernels of the hypothesis stage allocate from it (the
    first lists of the orientation phase, the followed lists), with full wavefronts and with 8 working lanes each."""
    s = host.Synth(1)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    monkeypatch.setenv("EG3D_ARENA_CAP0", "64")
    for lanes in ("64", "8"):
        monkeypatch.setenv("EG3D_K3A_ENGINE_LANES", lanes)
        ctx = api.Context(s.scene)
        got = ctx.match_refpoints(s.seeds)
        ctx.close()
        assert compare_edgepoints(ref, got, rel_tol=1e-4)["ok"]


def test_hypothesis_stage_with_few_working_lanes_looks_ahead(monkeypatch):
    """With few working lanes per wavefront every lane gets several of the wave's request slots: the orientation
    rounds, the replays and the followed lists run ahead of their answers (eg3d_k3a_engine.h). The result must not
    depend on it: 1, 2, 5 and 64 working lanes give the cloud of the oracle, flags included."""
    s = host.Synth(1)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    for lanes in ("1", "2", "5", "64"):
        monkeypatch.setenv("EG3D_K3A_ENGINE_LANES", lanes)
        ctx = api.Context(s.scene)
        got = ctx.match_refpoints(s.seeds)
        ctx.close()
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"], (lanes, rep["msgs"][:3])
        assert got["flags"] == ref["flags"], (lanes, got["flags"], ref["flags"])


def test_hypothesis_list_capacity_is_reported_not_truncated(monkeypatch):
    """A following direction of the hypothesis stage holds at most hyp_cap points (160 by default; the reference's
    vectors have no limit). A list that would outgrow it — its probe past the last point that fits succeeds — must
    make the call fail with EG3D_FLAG_HYP_OVERFLOW rather than return a truncated cloud; a capacity the longest list
    just fits must change nothing. Checked with every look-ahead depth (working lanes per wavefront)."""
    s = host.Synth(1)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    for lanes in ("64", "2"):
        monkeypatch.setenv("EG3D_K3A_ENGINE_LANES", lanes)
        monkeypatch.setenv("EG3D_HYP_CAP", "3")
        ctx = api.Context(s.scene)
        with pytest.raises(RuntimeError) as ei:
            ctx.match_refpoints(s.seeds)
        assert "capacity" in str(ei.value)
        ctx.close()
        # the smallest capacity that passes is the same whatever the look-ahead: bisect it, then check the cloud there
        lo, hi = 3, 160
        while hi - lo > 1:
            mid = (lo + hi) // 2
            monkeypatch.setenv("EG3D_HYP_CAP", str(mid))
            ctx = api.Context(s.scene)
            try:
                ctx.match_refpoints(s.seeds)
                hi = mid
            except RuntimeError:
                lo = mid
            ctx.close()
        monkeypatch.setenv("EG3D_HYP_CAP", str(hi))
        ctx = api.Context(s.scene)
        got = ctx.match_refpoints(s.seeds)
        ctx.close()
        assert compare_edgepoints(ref, got)["ok"]
        if lanes == "64":
            first = hi
        else:
            assert hi == first, (hi, first)
    monkeypatch.delenv("EG3D_HYP_CAP")


def test_slot_pool_size_does_not_matter_and_starvation_fails_loudly(monkeypatch):
    """The working slices of the expand stage are slots of a fixed arena, taken per XCD from a ring of slot ids
    (eg3d_kernels.hip pool_pop / pool_push). More slots than waves can be resident change nothing; a pool SMALLER than
    the residency (forced: EG3D_SLOTS_PER_XCD, a test knob) makes the waves queue for the slots of their XCD: the
    cloud is still complete, or — if a wave exhausts its bounded wait — the call fails with an error; never a hang,
    never a cloud with chains missing."""
    s = host.Synth(1)
    ref = _oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads=8)
    monkeypatch.setenv("EG3D_SLOTS_PER_XCD", "3000")
    ctx = api.Context(s.scene)
    got = ctx.match_refpoints(s.seeds)
    ctx.close()
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"], rep["msgs"][:3]
    monkeypatch.setenv("EG3D_SLOTS_PER_XCD", "2")
    ctx = api.Context(s.scene)
    try:  # the waves queue for the two slots of their XCD: either every chain gets its turn, or a wave gives up
        got = ctx.match_refpoints(s.seeds)
        rep = compare_edgepoints(ref, got)
        assert rep["ok"] and rep["bitexact_X"], rep["msgs"][:3]
    except RuntimeError as e:
        assert "working slice" in str(e)
    ctx.close()


def test_chain_expansion_in_many_chunks(monkeypatch):
    """Chains are expanded in chunks bounded by the scratch budget (24 GB by default); with a 48 MB
    budget config 1 needs several chunks — the concatenated output must not change."""
    s = host.Synth(1)
    ctx = api.Context(s.scene)
    whole = ctx.match_refpoints(s.seeds)
    ctx.close()
    monkeypatch.setenv("EG3D_MAX_SCRATCH_MB", "48")     # the knobs are read once, when a context is created
    ctx = api.Context(s.scene)
    monkeypatch.delenv("EG3D_MAX_SCRATCH_MB")
    parts = ctx.match_refpoints(s.seeds)
    assert not ctx.last_device_output().complete        # host-copy calls reuse the device buffers per chunk
    rep = compare_edgepoints(whole, parts)
    assert rep["ok"] and rep["bitexact_X"], rep["msgs"]
    # a DEVICE-ONLY call keeps the whole cloud in HBM however many chunks it took (what the RCCL gather reads)
    ctx.match_resident(0, s.n_seeds, device_only=True)
    dev = ctx.fetch_device_output()
    for k in ("X", "obs_xy"):
        assert np.array_equal(dev[k].view(np.uint32), whole[k].view(np.uint32)), k
    for k in ("obs_off", "key", "obs_view", "obs_pl", "obs_seg"):
        assert np.array_equal(dev[k], whole[k]), k
    # the same for the polyline-set path (key[0] = sample index of the whole call, added on the device)
    n_sets, row_off, ids = s.polyline_sets(3)
    sets_whole = ctx.match_polyline_sets(n_sets, row_off, ids)
    ctx.match_polyline_sets(n_sets, row_off, ids, device_only=True)
    dev = ctx.fetch_device_output()
    for k in ("obs_off", "key", "obs_view", "obs_pl", "obs_seg"):
        assert np.array_equal(dev[k], sets_whole[k]), k
    assert np.array_equal(dev["X"].view(np.uint32), sets_whole["X"].view(np.uint32))
    ctx.close()


def _read_match_sets(path):
    """A polyline match file of the example (text): returns (n_sets, row_off, ids)."""
    tok = open(path).read().split()
    assert tok[0] == "eg3d-polyline-sets" and tok[1] == "1"
    n_sets, n_views = int(tok[2]), int(tok[3])
    row_off, ids, k = [0], [], 4
    for _ in range(n_sets * n_views):
        c = int(tok[k])
        ids.extend(int(t) for t in tok[k + 1:k + 1 + c])
        k += 1 + c
        row_off.append(len(ids))
    return n_sets, np.asarray(row_off, np.uint32), np.asarray(ids, np.uint32)


def _concat_clouds(parts):
    """Clouds (dicts in the C ABI's layout) back to back, observation offsets rebased."""
    out = {"X": np.concatenate([p["X"].reshape(-1, 3) for p in parts]),
           "key": np.concatenate([p["key"].reshape(-1, 4) for p in parts]),
           "obs_view": np.concatenate([p["obs_view"] for p in parts]),
           "obs_pl": np.concatenate([p["obs_pl"] for p in parts]),
           "obs_seg": np.concatenate([p["obs_seg"] for p in parts]),
           "obs_xy": np.concatenate([p["obs_xy"].reshape(-1, 2) for p in parts])}
    off, base = [np.zeros(1, np.uint64)], 0
    for p in parts:
        off.append(p["obs_off"][1:].astype(np.uint64) + np.uint64(base))
        base += int(p["n_obs"])
    out["obs_off"] = np.concatenate(off)
    out["n_points"] = sum(int(p["n_points"]) for p in parts)
    out["n_obs"] = base
    return out


def _check_example_outputs_against_oracle(d, doc, doc_f, sets_files=()):
    from oracle import binding as ob
    H = host.lib()
    H.eg3d_sfm_read_json.restype = C.c_void_p
    H.eg3d_sfm_read_json.argtypes = [C.c_char_p]
    H.eg3d_sfm_destroy.argtypes = [C.c_void_p]
    H.eg3d_sfm_n_views.argtypes = [C.c_void_p]
    H.eg3d_sfm_cam_P.argtypes = [C.c_void_p]
    H.eg3d_sfm_cam_P.restype = D.f32p
    H.eg3d_sfm_seeds.argtypes = [C.c_void_p, C.POINTER(D.Seeds)]
    H.eg3d_sfm_points.argtypes = [C.c_void_p]
    H.eg3d_sfm_points.restype = D.f32p
    H.eg3d_sfm_analytic_F.argtypes = [C.c_void_p, D.f64p, D.u8p]
    H.eg3d_plg_read.restype = C.c_void_p
    H.eg3d_plg_read.argtypes = [C.c_char_p]
    H.eg3d_plg_scene.restype = C.POINTER(D.Scene)
    H.eg3d_plg_scene.argtypes = [C.c_void_p]
    H.eg3d_plg_destroy.argtypes = [C.c_void_p]
    sfm = H.eg3d_sfm_read_json((d + "/input.json").encode())
    plg = H.eg3d_plg_read((d + "/plgs.bin").encode())
    assert sfm and plg
    V = H.eg3d_sfm_n_views(sfm)
    F = np.zeros((V, V, 9), np.float64)
    Fv = np.zeros((V, V), np.uint8)
    assert H.eg3d_sfm_analytic_F(sfm, D.np_ptr(F, C.c_double), D.np_ptr(Fv, C.c_uint8)) == 0   # --all-pairs
    sc = D.Scene()
    C.memmove(C.byref(sc), H.eg3d_plg_scene(plg), C.sizeof(D.Scene))
    sc.cam_P = H.eg3d_sfm_cam_P(sfm)
    sc.F = D.np_ptr(F, C.c_double)
    sc.F_valid = D.np_ptr(Fv, C.c_uint8)
    seeds = D.Seeds()
    H.eg3d_sfm_seeds(sfm, C.byref(seeds))
    n0 = int(seeds.n_seeds)
    o = ob.Oracle(C.byref(sc))
    ref = o.match(C.byref(seeds), 0, n0, os.cpu_count())
    if sets_files:
        # pipelines 1 and 2 first, in stage order, then pipeline 3 (pipelines.cpp:219-227)
        stages = []
        for path in sets_files:
            ns, row_off, ids = _read_match_sets(path)
            stages.append(o.match_polyline_sets(ns, row_off, ids, 0, ns, os.cpu_count()))
            assert stages[-1]["n_points"] > 0
        ref = _concat_clouds(stages + [ref])
    ep = D.EdgePointsArrays(ref)
    keep = np.zeros(max(1, ref["n_points"]), np.uint8)
    assert ob.lib().orc_filter_close_2d(o._h, C.byref(ep.c), D.np_ptr(keep, C.c_uint8)) == 0
    # the expected SfM structure: the input's points, then the kept edge-points in emission order
    trk_off = D.as_np(seeds.trk_off, n0 + 1, np.uint32).astype(np.int64)
    n_trk = int(trk_off[-1])
    X = [D.as_np(H.eg3d_sfm_points(sfm), 3 * n0, np.float32).reshape(n0, 3).copy()]
    views = [D.as_np(seeds.trk_view, n_trk, np.int32).copy()]
    xy = [D.as_np(seeds.trk_xy, 2 * n_trk, np.float32).reshape(n_trk, 2).copy()]
    off = list(trk_off)
    eo = ref["obs_off"].astype(np.int64)
    for i in np.nonzero(keep[:ref["n_points"]])[0]:
        X.append(ref["X"][i:i + 1])
        views.append(ref["obs_view"][eo[i]:eo[i + 1]])
        xy.append(ref["obs_xy"][eo[i]:eo[i + 1]])
        off.append(off[-1] + int(eo[i + 1] - eo[i]))
    X, views, xy, off = np.concatenate(X), np.concatenate(views), np.concatenate(xy), np.array(off, np.int64)

    def same_structure(doc_structure, X, off, views, xy):
        assert len(doc_structure) == len(off) - 1
        for i, p in enumerate(doc_structure):
            v = p["value"]
            assert np.array_equal(np.array(v["X"], np.float64).astype(np.float32).view(np.uint32), X[i].view(np.uint32)), i
            ob_ = v["observations"]
            assert [q["key"] for q in ob_] == list(views[off[i]:off[i + 1]]), i
            got_xy = np.array([q["value"]["x"] for q in ob_], np.float64).astype(np.float32).reshape(-1, 2)
            assert np.array_equal(got_xy.view(np.uint32), xy[off[i]:off[i + 1]].view(np.uint32)), i
    same_structure(doc["structure"], X, off, views, xy)
    # ---- `--filter`: Gauss-Newton refinement of EVERY point (inliers moved), then the observation-count filter
    # on the edge-points, then removal
    off32 = off.astype(np.uint32)
    Xo, inl = o.gn_filter(X, off32, views, xy, 2.25, nthreads=os.cpu_count())
    Xf = np.where(inl[:, None] != 0, Xo, X)
    inl = np.ascontiguousarray(inl, np.uint8)
    ob.lib().orc_observation_filter(V, D.np_ptr(off32, C.c_uint32), len(off) - 1, n0, -1, D.np_ptr(inl, C.c_uint8))
    sel = np.nonzero(inl)[0]
    foff = np.concatenate([[0], np.cumsum((off[1:] - off[:-1])[sel])])
    fviews = np.concatenate([views[off[i]:off[i + 1]] for i in sel])
    fxy = np.concatenate([xy[off[i]:off[i + 1]] for i in sel])
    same_structure(doc_f["structure"], Xf[sel], foff, fviews, fxy)
    H.eg3d_plg_destroy(plg)
    H.eg3d_sfm_destroy(sfm)


def test_cpp_end_to_end_example(tmp_path):
    """examples/edge_matcher_refpoints.cpp: OpenMVG JSON + polyline-graph file in, pipeline 3 on the GPU
    through the C ABI, dedup, add points, (optional) ./filter -e, OpenMVG JSON out — all from C++.
    Its counts must equal the same steps driven from Python on the same synthetic scene."""
    import json
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "edgegraph3d_amd")
    exe = str(tmp_path / "edge_matcher_refpoints")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "edge_matcher_refpoints.cpp"), "-L", pkg, "-leg3d", "-leg3d_host",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib", "-lamdhip64", "-o", exe])
    d = str(tmp_path)
    subprocess.check_call([exe, "--make-synthetic", "1", d])
    out = subprocess.run([exe, d + "/input.json", d + "/plgs.bin", d + "/out.json", "--all-pairs"], capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"-> (\d+) edge-points \((\d+) observations\)", out.stdout)
    k = re.search(r"kept (\d+) edge-points", out.stdout)
    s = host.Synth(1)
    ctx = api.Context(s.scene)
    got = ctx.match_refpoints(s.seeds)
    ctx.close()
    assert int(m.group(1)) == got["n_points"] and int(m.group(2)) == got["n_obs"]
    doc = json.load(open(d + "/out.json"))
    assert len(doc["structure"]) == s.n_seeds + int(k.group(1)) and 0 < int(k.group(1)) < got["n_points"]
    assert len(doc["views"]) == s.n_views == len(doc["extrinsics"])
    # the appended points carry >= 3 observations with view keys inside the rig
    last = doc["structure"][-1]["value"]["observations"]
    assert len(last) >= 3 and all(0 <= o["key"] < s.n_views for o in last)
    out2 = subprocess.run([exe, d + "/input.json", d + "/plgs.bin", d + "/out_f.json", "--filter", "--all-pairs"],
                          capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    f = re.search(r"filter: (\d+) of (\d+) points kept", out2.stdout)
    assert int(f.group(2)) == len(doc["structure"]) and 0 < int(f.group(1)) <= int(f.group(2))
    assert len(json.load(open(d + "/out_f.json"))["structure"]) == int(f.group(1))
    # ---- N3 END TO END AGAINST THE ORACLE: the structure the example wrote (SfM points + de-duplicated edge-points,
    # and after `--filter` the Gauss-Newton-refined, observation-filtered set) must be what the ORACLE derives from
    # the same input files: oracle match -> orc_filter_close_2d -> append -> [orc_gn_filter -> orc_observation_filter
    # -> remove] (filtering_close_plgps.cpp:99-124, output_utilities.cpp:96-111, gauss_newton.cpp:136-178,
    # outliers_filtering.cpp:14-114). Only the file readers are the product's here.
    _check_example_outputs_against_oracle(d, doc, json.load(open(d + "/out_f.json")))
    # ---- ALL THREE STAGES of edge_reconstruction_pipeline in one run (pipelines.cpp:201-246): polyline matches of
    # pipelines 1 and 2 from files -> the extractor, then the reference points, concatenated in that order, ONE
    # de-duplication over the concatenation, append, filter — against the same chain made with the oracle
    for name, extra in (("out_3", []), ("out_3f", ["--filter"])):
        o3 = subprocess.run([exe, d + "/input.json", d + "/plgs.bin", d + "/%s.json" % name, "--all-pairs", "--sets1", d + "/sets1.txt",
                             "--sets2", d + "/sets2.txt"] + extra, capture_output=True, text=True, timeout=300)
        assert o3.returncode == 0, o3.stdout + o3.stderr
        assert re.search(r"pipeline 1: 3 polyline matches -> (\d+) edge-points", o3.stdout), o3.stdout
        assert re.search(r"pipeline 2: 3 polyline matches -> (\d+) edge-points", o3.stdout), o3.stdout
    doc3 = json.load(open(d + "/out_3.json"))
    assert len(doc3["structure"]) > len(doc["structure"])     # the two extra stages contributed points
    _check_example_outputs_against_oracle(d, doc3, json.load(open(d + "/out_3f.json")), (d + "/sets1.txt", d + "/sets2.txt"))
    # N4: the reference's rule for which view pairs have a fundamental matrix (>= 10 common SfM points) — the
    # default — must give what the library gives on the same scene with those pairs switched off; and the
    # matrices estimated from the tracks (--estimate-F) must reproduce most of the cloud
    off, view, xy = s.seeds_np()
    _, rule, _, _ = host.estimate_F(s.n_views, off, view, xy, estimate=False)
    sc = s.scene_np()
    sc["F_valid"] = (sc["F_valid"] & rule).astype(np.uint8)
    sa = host.SceneArrays(sc)
    ctx = api.Context(C.byref(sa.c))
    ruled = ctx.match_refpoints(s.seeds)
    ctx.close()
    out3 = subprocess.run([exe, d + "/input.json", d + "/plgs.bin", d + "/out_r.json"], capture_output=True, text=True,
                          timeout=300)
    assert out3.returncode == 0, out3.stdout + out3.stderr
    m3 = re.search(r"-> (\d+) edge-points \((\d+) observations\)", out3.stdout)
    assert int(m3.group(1)) == ruled["n_points"] and int(m3.group(2)) == ruled["n_obs"]
    out4 = subprocess.run([exe, d + "/input.json", d + "/plgs.bin", d + "/out_e.json", "--estimate-F"], capture_output=True,
                          text=True, timeout=300)
    assert out4.returncode == 0, out4.stdout + out4.stderr
    m4 = re.search(r"-> (\d+) edge-points", out4.stdout)
    assert 0.5 * ruled["n_points"] <= int(m4.group(1)) <= 1.5 * ruled["n_points"] + 10, out4.stdout


def test_invalid_polyline_that_kept_its_vertices_is_ignored():
    """include/eg3d.h allows pl_valid == 0 on a polyline whose vertex slice is not empty (the reference
    clears the coordinates of an invalidated polyline, polyline_graph_2d.cpp:1047-1058). Such
    vertices must be invisible to every path — including the polyline-sets path, which takes raw
    polyline ids: the result equals the oracle's and equals the scene with the slice removed."""
    s = host.Synth(1)
    sc = s.scene_np()
    n_sets, row_off, ids = s.polyline_sets(3)
    victims = [int(i) for i in ids[:4]]                      # view-0 polylines of the first set
    sc["pl_valid"] = sc["pl_valid"].copy()
    v0 = int(sc["view_pl_off"][0])
    for p in victims:
        assert sc["pl_vtx_off"][v0 + p + 1] - sc["pl_vtx_off"][v0 + p] >= 2
        sc["pl_valid"][v0 + p] = 0                           # invalid, vertices still there
    sa = host.SceneArrays(sc)
    ctx = api.Context(C.byref(sa.c))
    got = ctx.match_polyline_sets(n_sets, row_off, ids)
    ref = _oracle(C.byref(sa.c)).match_polyline_sets(n_sets, row_off, ids, nthreads=8)
    rep = compare_edgepoints(ref, got)
    assert rep["ok"] and rep["bitexact_X"], rep["msgs"]
    assert not np.isin(got["obs_pl"][got["obs_view"] == 0], victims).any()
    seeds_got = ctx.match_refpoints(s.seeds)
    seeds_ref = _oracle(C.byref(sa.c)).match(s.seeds, 0, s.n_seeds, nthreads=8)
    assert compare_edgepoints(seeds_ref, seeds_got)["ok"]
    ctx.close()


def test_inconsistent_seed_arrays_are_refused():
    """eg3d_upload_seeds: a descending or non-zero-based offset array and an out-of-range view id are refused (the
    kernels would index the camera and fundamental matrices with them)."""
    s = host.Synth(0)
    off, view, xy = s.seeds_np()
    ctx = api.Context(s.scene)
    bad_off = off.copy()
    bad_off[2] = bad_off[4] + 3
    for o, v in ((bad_off, view), (off + 1, view), (off, np.where(np.arange(len(view)) == 5, 99, view))):
        sd = host.SeedsArrays(o, v, xy)
        with pytest.raises(RuntimeError):
            ctx.upload_seeds(C.byref(sd.c))
    ctx.upload_seeds(s.seeds)                      # the context is still usable
    assert ctx.match_resident(0, s.n_seeds)["n_points"] > 0
    ctx.close()


def test_gn_filter_refuses_inconsistent_observation_arrays():
    s = host.Synth(0)
    X, off, view, xy = s.points(100)
    ctx = api.Context(s.scene)
    bad = off.copy()
    bad[10] = bad[20] + 5
    for o, v in ((bad, view), (off + 2, view), (off, np.where(np.arange(len(view)) == 3, -1, view))):
        with pytest.raises(RuntimeError):
            ctx.gn_filter(X, o, v, xy, 2.25)
    Xo, inl, _ = ctx.gn_filter(X, off, view, xy, 2.25)
    assert inl.sum() > 0
    ctx.close()
