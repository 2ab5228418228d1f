#!/usr/bin/env python3
"""GPU box: what the FIRST eg3d_match_resident call on a fresh context costs (work buffers, lanes, pinned staging are
allocated inside it) against a warm one, per pipelining setting and output mode. usage: cold_call_probe.py <c2|c3|c4>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from edgegraph3d_amd import api, host  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
s = host.Synth({"c2": 2, "c3": 3, "c4": 4}[wl])
n = s.n_seeds if wl != "c4" else 8192
api.Context(s.scene).close()  # HIP runtime start-up is not what is measured
for lanes, dev in ((1, True), (1, False), (3, False), (0, False), (3, True)):
    t0 = time.perf_counter()
    ctx = api.Context(s.scene)
    t1 = time.perf_counter()
    ctx.upload_seeds(s.seeds)
    ctx.set_pipelining(lanes, 0)
    t2 = time.perf_counter()
    if dev:
        r = ctx.match_resident(0, n, device_only=True)
        cold = time.perf_counter() - t2
        print("   cold stage times:", {k: round(v, 2) for k, v in r["times"].items() if k.startswith("ms_")}, flush=True)
        t3 = time.perf_counter()
        ctx.match_resident(0, n, device_only=True)
        warm = time.perf_counter() - t3
    else:
        cold, _ = ctx.time_match_to_host(0, n)
        warm, _ = ctx.time_match_to_host(0, n)
    print("%s lanes %d %-11s create %.1f ms  first call %.1f ms  second call %.1f ms  (cold extra %.1f ms)"
          % (wl, lanes, "device-only" if dev else "to host", (t1 - t0) * 1e3, cold * 1e3, warm * 1e3, (cold - warm) * 1e3), flush=True)
    ctx.close()
