#!/usr/bin/env python3
"""GPU box, under `rocprofv3 --kernel-trace`: a few eg3d_match_resident calls per pipelining setting, separated by sleeps,
so that the kernel trace shows how the units' kernels overlap. usage: timeline_probe.py <wl> <lanes>x<units>[,..] [host]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from edgegraph3d_amd import api, host  # noqa: E402

wl = sys.argv[1]
settings = [tuple(int(x) for x in p.split("x")) for p in sys.argv[2].split(",")]
to_host = len(sys.argv) > 3 and sys.argv[3] == "host"
s = host.Synth({"c2": 2, "c3": 3, "c4": 4}[wl])
n = s.n_seeds if wl != "c4" else 8192
ctx = api.Context(s.scene)
ctx.upload_seeds(s.seeds)
for lanes, units in settings:
    ctx.set_pipelining(lanes, units)
    for _ in range(2):
        ctx.match_resident(0, n, device_only=True)
        if to_host:
            ctx.time_match_to_host(0, n)
    time.sleep(0.2)
    for rep in range(2):
        t0 = time.perf_counter()
        if to_host:
            ctx.time_match_to_host(0, n)
        else:
            ctx.match_resident(0, n, device_only=True)
        print("CALL %dx%d rep %d: %.3f ms" % (lanes, units, rep, 1e3 * (time.perf_counter() - t0)), flush=True)
        time.sleep(0.2)
ctx.close()
