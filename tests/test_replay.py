"""Row a17: the host replay of PLGMatchesManager::add_matched_3dpolyline (product, flat arrays)
against the oracle's reference-shaped restatement (hash map / std::set), plus hand-built cases for
the quirks of the structure: exact-coordinate node identity, duplicate suppression in either
orientation, the one-interval-per-start-segment std::set, the cross-polyline extreme rule and the
-1 "invalid coordinate" node."""
import numpy as np
import pytest

from edgegraph3d_amd import host
from oracle import binding as ob

FIELDS = ("n_nodes", "n_real_nodes", "n_polylines")
ARRAYS = ("node_X", "node_point", "pl_start", "pl_end", "conn_off", "conn_pl", "iv_off", "iv_start_seg", "iv_start_xy",
          "iv_end_seg", "iv_end_xy")


def _same(a, b):
    for f in FIELDS:
        assert a[f] == b[f], f
    for f in ARRAYS:
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f
        else:
            assert np.array_equal(x, y), f


def _cloud(points):
    """points = list of (X, [(view, pl, seg, x, y)...], key4)"""
    X = np.array([p[0] for p in points], np.float32).reshape(-1, 3)
    off, view, pl, seg, xy, key = [0], [], [], [], [], []
    for _, obs, k in points:
        for (v, p, s, x, y) in obs:
            view.append(v), pl.append(p), seg.append(s), xy.append((x, y))
        off.append(len(view))
        key.append(k)
    return {"X": X, "obs_off": np.array(off, np.uint32), "obs_view": np.array(view, np.int32),
            "obs_pl": np.array(pl, np.uint32), "obs_seg": np.array(seg, np.uint32),
            "obs_xy": np.array(xy, np.float32).reshape(-1, 2), "key": np.array(key, np.uint32).reshape(-1, 4)}


def test_replay_matches_oracle_on_synthetic_clouds():
    for cfg in (0, 1):
        s = host.Synth(cfg)
        o = ob.Oracle(s.scene)
        cloud = o.match(s.seeds, 0, s.n_seeds, 1)
        assert cloud["n_points"] > 100
        got = host.replay_matches(s.scene, cloud)
        ref = o.replay_matches(cloud)
        _same(got, ref)
        assert got["n_polylines"] > 0 and got["iv_off"][-1] > 0
        # every chain of L points adds at most L-1 connections, and nodes are shared by coordinates
        assert got["n_polylines"] <= cloud["n_points"] - len(np.unique(cloud["key"][:, :3], axis=0))
        assert got["n_nodes"] <= cloud["n_points"]


def test_replay_quirks_hand_built():
    s = host.Synth(0)
    sc = s.scene_np()
    o = ob.Oracle(s.scene)
    # pick a polyline of view 0 with >= 4 vertices, and a neighbour sharing its end node if any
    vpo, pvo = sc["view_pl_off"], sc["pl_vtx_off"]
    n = np.diff(pvo)
    pl0 = next(p for p in range(int(vpo[0]), int(vpo[1])) if n[p] >= 4)
    v = sc["vtx_xy"][int(pvo[pl0]):int(pvo[pl0 + 1])]
    lp = pl0 - int(vpo[0])
    mid = lambda a, b, t: (float(a[0] + t * (b[0] - a[0])), float(a[1] + t * (b[1] - a[1])))
    A, B, Cc = (1.0, 2.0, 3.0), (4.0, 5.0, 6.0), (7.0, 8.0, 9.0)
    pts = [
        # chain 0: A -> B -> A (the reverse connection is a duplicate) -> C
        (A, [(0, lp, 0, *mid(v[0], v[1], 0.25))], (0, 0, 0, 0)),
        (B, [(0, lp, 0, *mid(v[0], v[1], 0.75))], (0, 0, 0, 1)),      # same segment: ordered by position
        (A, [(0, lp, 1, *mid(v[1], v[2], 0.5))], (0, 0, 0, 2)),       # interval (seg0 .. seg1): start seg 0 again -> dropped
        (Cc, [(0, lp, 2, *mid(v[2], v[3], 0.5))], (0, 0, 0, 3)),
        # chain 1 (new key): -0.0 equals +0.0, NaN never equals, x == -1 is an "invalid" node
        ((0.0, 1.0, 1.0), [(0, lp, 0, *mid(v[0], v[1], 0.5))], (1, 0, 0, 0)),
        ((-0.0, 1.0, 1.0), [(0, lp, 1, *mid(v[1], v[2], 0.5))], (1, 0, 0, 1)),   # loop polyline on one node
        ((float("nan"), 0.0, 0.0), [], (1, 0, 0, 2)),
        ((float("nan"), 0.0, 0.0), [], (1, 0, 0, 3)),
        ((-1.0, 2.0, 2.0), [], (1, 0, 0, 4)),
        ((-1.0, 2.0, 2.0), [], (1, 0, 0, 5)),
        # key[3] not consecutive -> a new chain even with the same (seed, entry, hit)
        (B, [], (1, 0, 0, 9)),
    ]
    cloud = _cloud(pts)
    got = host.replay_matches(s.scene, cloud)
    ref = o.replay_matches(cloud)
    _same(got, ref)
    # nodes (every segment looks both of its points up): A0 B1 C2 | zero3 (-0 == +0: the same node) |
    # NaN never matches: 4, then 5 and 6, then 7 | x == -1 is an "invalid" node: 8 is created, found
    # again and wiped for 9, which is found again and wiped for 10
    assert got["n_nodes"] == 11 and got["n_real_nodes"] == 11
    # polylines: A-B, (B-A is the duplicate), A-C, the zero loop, ...
    assert [tuple(x) for x in np.stack([got["pl_start"], got["pl_end"]], 1)][:4] == [(0, 1), (0, 2), (3, 3), (3, 4)]
    for wiped in (8, 9):
        assert np.array_equal(got["node_X"][wiped], np.array([-1, -1, -1], np.float32))
    assert got["conn_off"][9] == got["conn_off"][8]  # invalidate_node cleared node 8's connections
    # view 0 / polyline lp: the set keeps ONE interval starting on segment 0 (the first inserted) and one on 1
    a, b = int(got["iv_off"][pl0]), int(got["iv_off"][pl0 + 1])
    assert list(got["iv_start_seg"][a:b]) == [0, 1]
    assert got["iv_end_seg"][a] == 0  # A..B on the same segment, not the later 0..1 interval


def test_replay_refuses_a_cloud_whose_ids_do_not_fit_the_scene():
    s = host.Synth(0)
    cloud = ob.Oracle(s.scene).match(s.seeds, 0, s.n_seeds, 1)
    V = s.scene_np()["n_views"]
    for field, value in (("obs_view", V), ("obs_view", -1), ("obs_pl", 10**6), ("obs_seg", 10**6)):
        bad = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in cloud.items()}
        bad[field][5] = value
        with pytest.raises(Exception):
            host.replay_matches(s.scene, bad)
    bad = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in cloud.items()}
    bad["obs_off"][3] = bad["obs_off"][6] + 1
    with pytest.raises(Exception):
        host.replay_matches(s.scene, bad)
    assert host.replay_matches(s.scene, cloud)["n_nodes"] > 0
