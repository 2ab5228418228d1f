#!/usr/bin/env python3
"""Per-section breakdown of k3b_expand on a C4-shaped subset (200 views); needs the timing build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegraph3d_amd import api, host
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
cfg = host.default_config(4); cfg.n_seeds = n
s = host.Synth(cfg); ctx = api.Context(s.scene); ctx.upload_seeds(s.seeds)
for _ in range(2):
    r = ctx.match_resident(0, n, device_only=True)
L = api.lib()
L.eg3d_probe_sections.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
sm, sl, nc = (C.c_double * 16)(), (C.c_double * 16)(), C.c_uint32()
assert L.eg3d_probe_sections(ctx._h, sm, sl, C.byref(nc)) == 0
names = ["candidates", "stepwalks", "sidewalks", "batchGN", "follow", "stepDLT", "stepGN", "whole", "commit", "init", "epc-pre", "newpoint"]
print("chains", nc.value, "points", r["n_points"], "obs", r["n_obs"], "ms_expand", r["times"]["ms_expand"])
for k in range(12):
    print("%-11s sum %10.3e (%5.1f%%)  slowest %10.3e" % (names[k], sm[k], 100 * sm[k] / (sm[7] or 1), sl[k]))
