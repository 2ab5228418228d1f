#!/usr/bin/env python3
"""GPU box: parity of ONE WHOLE bench step of C4 (8192 seeds, ~4.65 M edge-points, 342 M observations) against the
oracle run on all host cores -> gpurun_out/c4_batch_parity.json. (The GPU tests check 1200 seeds; the bench's
parity leg 128.)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from edgegraph3d_amd import api, host  # noqa: E402
from oracle import binding as ob  # noqa: E402
from parity_util import compare_edgepoints  # noqa: E402
from bench import _usable_cores  # noqa: E402

NCORES = _usable_cores()[0]  # what the box grants (cgroup quota), not what it shows

# usage: c4_batch_parity.py [seeds per batch = 8192] [first batch = 0] [batches = 1]   (13 batches of 8192 = all 100 000 seeds)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s = host.Synth(4)
ctx = api.Context(s.scene)
ctx.upload_seeds(s.seeds)
orc = ob.Oracle(s.scene)
os.makedirs("gpurun_out", exist_ok=True)
rows = []
for b in range(first, first + count):
    lo, hi = b * n, min(s.n_seeds, (b + 1) * n)
    if lo >= hi:
        break
    t0 = time.time()
    got = ctx.match_resident(lo, hi)
    t1 = time.time()
    ref = orc.match(s.seeds, lo, hi, NCORES)
    t2 = time.time()
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    row = {"seeds": [lo, hi], "points": int(ref["n_points"]), "observations": int(ref["n_obs"]),
           "structure_exact": bool(rep["ok"]), "X_bit_exact": bool(rep.get("bitexact_X")),
           "obs_xy_bit_exact": bool(rep.get("bitexact_xy")), "max_rel_err_X": rep.get("max_rel_X"),
           "flags_device": int(got["flags"]), "flags_oracle": int(ref["flags"]),
           "device_seconds_incl_copy": round(t1 - t0, 2), "oracle_seconds_all_cores": round(t2 - t1, 1), "cores": NCORES}
    print(row, flush=True)
    rows.append(row)
    json.dump(rows, open("gpurun_out/c4_batch_parity_%d.json" % first, "w"), indent=1)
    del got, ref
    assert row["structure_exact"] and row["X_bit_exact"]
