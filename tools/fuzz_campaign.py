#!/usr/bin/env python3
"""GPU box: a longer run of the device fuzz test (tests/test_gpu_fuzz.py::test_random_mutated_scene_matches_oracle):
`salts` x 100 random mutated scenes, each compared with the oracle bit for bit — the seed path (cloud, flags, stage A
alone) and the polyline-set path. Prints one JSON summary; exit code 1 on any mismatch.
usage: python tools/fuzz_campaign.py <first salt> <n salts> [out.json] [rows: 3 = the default 6x4 DLT form, 2 = libeg3d_dlt4x4.so]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from edgegraph3d_amd import api  # noqa: E402
from fuzz_scenes import draw  # noqa: E402
from oracle import binding as ob  # noqa: E402
from parity_util import compare_edgepoints  # noqa: E402


def one(case, salt):
    s, sa, seeds = draw(case, salt)
    n = len(seeds.trk_off) - 1
    ctx = api.Context(C.byref(sa.c))
    orc = ob.Oracle(C.byref(sa.c))
    msgs = []
    got = ctx.match_refpoints(C.byref(seeds.c), 0, n)
    ref = orc.match(C.byref(seeds.c), 0, n, nthreads=16)
    rep = compare_edgepoints(ref, got)
    if not (rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"]):
        msgs.append("cloud: %s" % rep["msgs"][:2])
    if got["flags"] != ref["flags"]:
        msgs.append("flags %d != %d" % (got["flags"], ref["flags"]))
    ca, cb = ctx.candidates(C.byref(seeds.c), 0, n), orc.candidates(C.byref(seeds.c), 0, n)
    for k in cb:
        x, y = ca[k], cb[k]
        if isinstance(y, np.ndarray):
            bits = (lambda a: a.view(np.uint32) if a.dtype == np.float32 else a)
            if not np.array_equal(bits(x), bits(y)):
                msgs.append("stage A: " + k)
        elif x != y:
            msgs.append("stage A: " + k)
    n_sets, row_off, ids = s.polyline_sets(2)
    gs = ctx.match_polyline_sets(n_sets, row_off, ids)
    rs = orc.match_polyline_sets(n_sets, row_off, ids, 0, n_sets, 16)
    rep = compare_edgepoints(rs, gs)
    if not (rep["ok"] and rep["bitexact_X"] and rep["bitexact_xy"]):
        msgs.append("sets: %s" % rep["msgs"][:2])
    ctx.close()
    return got["n_points"], gs["n_points"], msgs


def main():
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    from forms import product_form
    with product_form(rows):
        return campaign()


def campaign():
    s0, ns = int(sys.argv[1]), int(sys.argv[2])
    out = sys.argv[3] if len(sys.argv) > 3 else None
    t0 = time.time()
    bad, pts, spts, cases = [], 0, 0, 0
    for salt in range(s0, s0 + ns):
        for case in range(100):
            a, b, msgs = one(case, salt)
            cases += 1
            pts += a
            spts += b
            if msgs:
                bad.append({"salt": salt, "case": case, "msgs": msgs})
                print("MISMATCH", salt, case, msgs, flush=True)
    summary = {"salts": [s0, s0 + ns], "scenes": cases, "edge_points_seed_path": int(pts), "edge_points_set_path": int(spts),
               "mismatches": bad, "seconds": round(time.time() - t0, 1), "dlt_form_rows": int(api.lib().eg3d_dlt_rows()),
               "what": "tests/fuzz_scenes.py draw(case, salt): random mutated scenes (3-27 views, invalid F pairs, loops, invalid "
                       "polylines, repeated views, border / outside seeds, cut tracks); device == oracle bit for bit: cloud, "
                       "flags, stage A (candidates, start hits, epipolar hits), polyline-set path"}
    print(json.dumps(summary))
    if out:
        json.dump(summary, open(out, "w"), indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
