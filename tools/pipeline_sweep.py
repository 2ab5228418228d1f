#!/usr/bin/env python3
"""GPU box: what ONE eg3d_match_resident call costs as a function of the internal pipelining (eg3d_set_pipelining lanes x
units): device-only wall time and end to end (with the D2H copy into caller-owned arrays), median of `reps` calls after
two warm-up calls per setting. usage: pipeline_sweep.py <c2|c3|c3real|c4> [reps] [batch seeds]  -> one JSON line."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from edgegraph3d_amd import api, host  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if wl == "c3real":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import real_scene as rs
    import ctypes as C
    sc, seeds, _ = rs.real_edges_scene(n_seeds=6268)
    scene, sd, n = C.pointer(sc.c), C.pointer(seeds.c), int(seeds.c.n_seeds)
else:
    s = host.Synth({"c2": 2, "c3": 3, "c4": 4}[wl])
    scene, sd, n = s.scene, s.seeds, s.n_seeds
if batch:
    n = min(n, batch)
elif wl == "c4":
    n = 8192
t0 = time.perf_counter()
ctx = api.Context(scene)
t_create = time.perf_counter() - t0
t0 = time.perf_counter()
ctx.upload_seeds(sd)
t_upload = time.perf_counter() - t0
settings = [(1, 0), (2, 0), (3, 0), (4, 0), (4, 6), (4, 8), (4, 12), (6, 6), (6, 12), (8, 8), (8, 16)]
if os.environ.get("SWEEP"):
    settings = [tuple(int(x) for x in p.split("x")) for p in os.environ["SWEEP"].split(",")]
rows = []
for lanes, units in settings:
    ctx.set_pipelining(lanes, units)
    for _ in range(2):
        ctx.match_resident(0, n, device_only=True)
        ctx.time_match_to_host(0, n)
    dev, e2e, pts = [], [], 0
    for _ in range(reps):
        t0 = time.perf_counter()
        r = ctx.match_resident(0, n, device_only=True)
        dev.append(time.perf_counter() - t0)
        pts = r["n_points"]
    for _ in range(reps):
        t, _n = ctx.time_match_to_host(0, n)
        e2e.append(t)
    rows.append({"lanes": lanes, "units": units, "device_only_ms": round(1e3 * statistics.median(dev), 3),
                 "device_only_min_ms": round(1e3 * min(dev), 3), "end_to_end_ms": round(1e3 * statistics.median(e2e), 3),
                 "end_to_end_min_ms": round(1e3 * min(e2e), 3)})
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps({"workload": wl, "seeds": n, "edge_points": pts, "create_ms": round(1e3 * t_create, 2),
                  "upload_seeds_ms": round(1e3 * t_upload, 2), "reps": reps, "rows": rows}))
