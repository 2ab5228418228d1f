#!/usr/bin/env python3
"""C4-shaped robustness check on the GPU: 200 views / ~20k segments per view; a subset of the
100k seeds is compared with the oracle (all cores), then a larger range is timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from edgegraph3d_amd import api, host
from oracle import binding as ob
from parity_util import compare_edgepoints

n_par = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_big = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cfg = host.default_config(4)
cfg.n_seeds = max(n_par, n_big)
t = time.time(); s = host.Synth(cfg); print("synth C4: views", s.n_views, "segs/view", s.total_segments / s.n_views, "gen s", round(time.time() - t, 1))
t = time.time(); ctx = api.Context(s.scene); print("eg3d_create (grids + upload) s", round(time.time() - t, 1))
ctx.upload_seeds(s.seeds)
got = ctx.match_resident(0, n_par)
print("gpu subset", got["n_points"], got["n_obs"], "flags", got["flags"], got["times"])
t = time.time(); o = ob.Oracle(s.scene); print("oracle create s", round(time.time() - t, 1))
ref = o.match(s.seeds, 0, n_par, os.cpu_count())
print("oracle subset", ref["n_points"], "cpu s (all cores)", round(ref["stats"]["seconds"], 2))
rep = compare_edgepoints(ref, got)
print("PARITY", rep["ok"], rep.get("msgs"), "bitexact", rep.get("bitexact_X"))
r = ctx.match_resident(0, n_big, device_only=True)
print("gpu big", n_big, "seeds:", r["n_points"], "points", r["n_obs"], "obs; chains", r["n_chains"], "hyp", r["n_hypotheses"], "flags", r["flags"])
print("times", r["times"]); print("edge-points/s", r["n_points"] / (r["times"]["ms_total"] * 1e-3))
