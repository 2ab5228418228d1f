#!/usr/bin/env python3
"""Generates the committed golden fixtures. The reference ships no golden vectors and cannot be
run here, so these are produced by the CPU oracle of this repo on seeded
synthetic scenes (inputs AND expected outputs are stored, so the fixture also pins the synthetic
generator). Regenerate with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from edgegraph3d_amd import host  # noqa: E402
from oracle import binding as ob  # noqa: E402


def make(cfg_index, name):
    s = host.Synth(cfg_index)
    o = ob.Oracle(s.scene)
    r = o.match(s.seeds, 0, s.n_seeds, 1)
    c = o.candidates(s.seeds, 0, s.n_seeds)
    sc = s.scene_np()
    off, view, xy = s.seeds_np()
    X, poff, pview, pxy = s.points(300)
    Xo, inl = o.gn_filter(X, poff, pview, pxy, 3.0)
    out = {"scene_" + k: v for k, v in sc.items()}
    out.update({"seeds_trk_off": off, "seeds_trk_view": view, "seeds_trk_xy": xy})
    for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key"):
        out["out_" + k] = r[k]
    out["out_counts"] = np.array([r["n_tasks"], r["stats"]["n_chains"], r["flags"]], np.int64)
    for k in ("cand_off", "cand_pl", "start_off", "start_pl", "start_seg", "start_xy", "task_sv", "task_hit",
              "task_list_off", "list_off", "hit_pl", "hit_seg", "hit_xy"):
        out["cand_" + k] = c[k]
    out.update({"gn_X": X, "gn_off": poff, "gn_view": pview, "gn_xy": pxy, "gn_Xout": Xo, "gn_inlier": inl})
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "points", r["n_points"], "obs", r["n_obs"])


def make_sets(cfg_index, name, max_sets):
    """Pipelines 1-2 extractor: the first `max_sets` synthetic polyline sets of the
    scene `cfg_index` (the scene itself is pinned by synthetic_tiny_v1.npz) and the oracle's output."""
    s = host.Synth(cfg_index)
    n, row_off, ids = s.polyline_sets(max_sets)
    r = ob.Oracle(s.scene).match_polyline_sets(n, row_off, ids, 0, n, 1)
    out = {"n_sets": np.int64(n), "row_off": row_off, "pl_ids": ids}
    for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key"):
        out["out_" + k] = r[k]
    out["out_counts"] = np.array([r["stats"]["n_tasks"], r["stats"]["n_chains"], r["flags"]], np.int64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "sets", n, "points", r["n_points"], "obs", r["n_obs"])


def make_dlt6x4(cfg_index, name):
    """Outputs of the same scene with the three-rows-per-view (6x4, OpenCV <= 3.1) DLT system; the
    inputs are those of synthetic_tiny_v1.npz."""
    assert ob.lib().orc_set_dlt_rows(3) == 0
    s = host.Synth(cfg_index)
    r = ob.Oracle(s.scene).match(s.seeds, 0, s.n_seeds, 1)
    assert ob.lib().orc_set_dlt_rows(2) == 0
    out = {"out_" + k: r[k] for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")}
    out["out_counts"] = np.array([r["n_tasks"], r["stats"]["n_chains"], r["flags"]], np.int64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "points", r["n_points"], "obs", r["n_obs"])


def make_sets_dlt6x4(cfg_index, name, max_sets):
    """Outputs of the sets of synthetic_tiny_sets_v1.npz with the 6x4 DLT system (outputs only)."""
    assert ob.lib().orc_set_dlt_rows(3) == 0
    s = host.Synth(cfg_index)
    n, row_off, ids = s.polyline_sets(max_sets)
    r = ob.Oracle(s.scene).match_polyline_sets(n, row_off, ids, 0, n, 1)
    assert ob.lib().orc_set_dlt_rows(2) == 0
    out = {"out_" + k: r[k] for k in ("X", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy", "key")}
    out["out_counts"] = np.array([r["stats"]["n_tasks"], r["stats"]["n_chains"], r["flags"]], np.int64)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "sets", n, "points", r["n_points"], "obs", r["n_obs"])


if __name__ == "__main__":
    # the v1 base files hold the inputs and the outputs of the 4x4 form; the *_dlt6x4 files the outputs of the 6x4
    # form (OpenCV <= 3.1, the product's default since round 3). `--only-new` leaves committed files untouched.
    os.environ["EG3D_ORACLE_DLT_ROWS"] = "2"
    only_new = "--only-new" in sys.argv

    def want(name):
        return not (only_new and os.path.exists(os.path.join(HERE, name)))
    if want("synthetic_tiny_v1.npz"):
        make(0, "synthetic_tiny_v1.npz")
    if want("synthetic_tiny_sets_v1.npz"):
        make_sets(0, "synthetic_tiny_sets_v1.npz", 3)
    if want("synthetic_tiny_v1_dlt6x4.npz"):
        make_dlt6x4(0, "synthetic_tiny_v1_dlt6x4.npz")
    if want("synthetic_tiny_sets_v1_dlt6x4.npz"):
        make_sets_dlt6x4(0, "synthetic_tiny_sets_v1_dlt6x4.npz", 3)
