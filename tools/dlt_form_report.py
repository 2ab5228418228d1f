#!/usr/bin/env python3
"""How much does the unpinned OpenCV version matter? Runs the CPU oracle on C2 and C3' with both
forms of cv::triangulatePoints' linear system (2 rows per view = 4x4, later OpenCV; 3 rows = 6x4,
OpenCV 2.4-3.1) and reports the seeds / points whose output differs structurally or by more than
1e-4 relative in X. CPU only (oracle = test infrastructure); writes profiles/r02_dlt_form_report.json."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgegraph3d_amd import host
from oracle import binding as ob

def run(cfg, rows, nthreads):
    assert ob.lib().orc_set_dlt_rows(rows) == 0
    s = host.Synth(cfg)
    return ob.Oracle(s.scene).match(s.seeds, 0, s.n_seeds, nthreads)

def chains(r):
    """dict: chain key (seed, entry, hit) -> (start, end) point range"""
    k = r["key"]
    out, b = {}, 0
    for i in range(1, len(k) + 1):
        if i == len(k) or tuple(k[i][:3]) != tuple(k[b][:3]):
            out[tuple(int(x) for x in k[b][:3])] = (b, i)
            b = i
    return out

report = {}
for name, cfg in (("C2", 2), ("C3'", 3)):
    a, b = run(cfg, 2, os.cpu_count()), run(cfg, 3, os.cpu_count())
    ca, cb = chains(a), chains(b)
    common = set(ca) & set(cb)
    same_struct, moved, maxrel = 0, 0, 0.0
    pts_compared = pts_moved = 0
    for key in common:
        (a0, a1), (b0, b1) = ca[key], cb[key]
        ok = (a1 - a0 == b1 - b0 and np.array_equal(np.diff(a["obs_off"][a0:a1 + 1]), np.diff(b["obs_off"][b0:b1 + 1])))
        if ok:
            oa, ob_ = a["obs_off"][a0], b["obs_off"][b0]
            n = a["obs_off"][a1] - oa
            ok = (np.array_equal(a["obs_view"][oa:oa + n], b["obs_view"][ob_:ob_ + n]) and
                  np.array_equal(a["obs_pl"][oa:oa + n], b["obs_pl"][ob_:ob_ + n]) and
                  np.array_equal(a["obs_seg"][oa:oa + n], b["obs_seg"][ob_:ob_ + n]))
        if not ok:
            continue
        same_struct += 1
        Xa, Xb = a["X"][a0:a1].astype(np.float64), b["X"][b0:b1].astype(np.float64)
        rel = np.linalg.norm(Xa - Xb, axis=1) / np.maximum(np.linalg.norm(Xa, axis=1), 1e-12)
        pts_compared += len(rel)
        pts_moved += int((rel > 1e-4).sum())
        maxrel = max(maxrel, float(rel.max()) if len(rel) else 0.0)
        moved += int((rel > 1e-4).any())
    report[name] = {
        "points_4x4": int(a["n_points"]), "points_6x4": int(b["n_points"]),
        "chains_4x4": len(ca), "chains_6x4": len(cb), "chains_in_both": len(common),
        "chains_only_4x4": len(set(ca) - set(cb)), "chains_only_6x4": len(set(cb) - set(ca)),
        "chains_structurally_identical": same_struct,
        "chains_identical_structure_but_X_moved_gt_1e-4": moved,
        "points_compared": pts_compared, "points_X_moved_gt_1e-4": pts_moved, "max_rel_dX": maxrel,
        "bit_identical_X_points": None,
    }
    print(name, json.dumps(report[name]))
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(report, open(os.path.join(ROOT, "profiles", "r02_dlt_form_report.json"), "w"), indent=1)
