#!/usr/bin/env python3
"""GPU box: narrow a device/oracle mismatch inside a seed range of a synthetic config down to one seed."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from edgegraph3d_amd import api, host  # noqa: E402
from oracle import binding as ob  # noqa: E402
from parity_util import compare_edgepoints  # noqa: E402

cfg, lo, hi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
s = host.Synth(cfg)
ctx = api.Context(s.scene)
ctx.upload_seeds(s.seeds)
orc = ob.Oracle(s.scene)


def bad(a, b):
    got = ctx.match_resident(a, b)
    ref = orc.match(s.seeds, a, b, os.cpu_count())
    rep = compare_edgepoints(ref, got, rel_tol=1e-4)
    return (not rep["ok"]), rep, ref, got


while hi - lo > 1:
    parts = 8 if hi - lo >= 64 else 2
    step = (hi - lo + parts - 1) // parts
    found = None
    for a in range(lo, hi, step):
        b = min(hi, a + step)
        isbad, rep, _, _ = bad(a, b)
        print("range", a, b, "BAD" if isbad else "ok", flush=True)
        if isbad:
            found = (a, b)
            break
    if not found:
        print("no failing sub-range: the mismatch needs the whole range", lo, hi)
        break
    lo, hi = found
isbad, rep, ref, got = bad(lo, hi)
print("seed range", lo, hi, "bad" if isbad else "ok")
print("messages:", rep["msgs"][:6])
print("ref points/obs", ref["n_points"], ref["n_obs"], "got", got["n_points"], got["n_obs"], "flags", ref["flags"], got["flags"])
off, view, xy = s.seeds_np()
print("track", view[off[lo]:off[lo + 1]].tolist())
for name in ("key", "obs_off", "obs_view", "obs_pl", "obs_seg"):
    r, g = ref[name], got[name]
    if r.shape != g.shape:
        print(name, "shape", r.shape, g.shape)
        continue
    d = np.nonzero((r != g).reshape(len(r), -1).any(1))[0]
    print(name, "differs at", d[:10].tolist(), "ref", r[d[:5]].tolist(), "got", g[d[:5]].tolist())
np.savez_compressed("gpurun_out/mismatch_seed.npz", lo=lo, hi=hi, **{"ref_" + k: ref[k] for k in ("X", "key", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy")},
                    **{"got_" + k: got[k] for k in ("X", "key", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy")})
