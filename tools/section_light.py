#!/usr/bin/env python3
"""Diagnostic: ONE section of k3b_expand's per-chain clocks, from a LIGHT timing build
(tools/build_variant.sh sec<k> -DEG3D_SECTION_TIMING -DEG3D_ONE_SECTION=<k>; EG3D_LIB selects the library): only section k
and the whole chain are accumulated, so the kernel keeps its speed (its HIP-event time is printed beside the shares: compare
with the product build's).  usage: EG3D_LIB=... python tools/section_light.py <k> [config]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegraph3d_amd import api, host  # noqa: E402

NAMES = ["candidates", "stepwalks", "sidewalks", "batchGN", "follow", "stepDLT", "stepGN", "whole", "commit", "init",
         "presolves", "newpoint", "expand_to_view", "attach_view", "fallback", "seq_steps"]
k = int(sys.argv[1])
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = host.Synth(cfg)
ctx = api.Context(s.scene)
ctx.upload_seeds(s.seeds)
ms = []
n_use = min(s.n_seeds, int(os.environ.get("EG3D_SECTION_SEEDS", "8192")))
for _ in range(4):
    r = ctx.match_resident(0, n_use, device_only=True)
    ms.append(r["times"]["ms_expand"])
L = api.lib()
L.eg3d_probe_sections_raw.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
sm, n = (C.c_double * 16)(), C.c_uint32()
assert L.eg3d_probe_sections_raw(ctx._h, sm, C.byref(n)) == 0
print("section %2d %-15s share of chain clocks %6.2f %%   (ticks %.4e of %.4e, %d chains; k3b_expand %.2f ms, runs %s)" % (
    k, NAMES[k], 100.0 * sm[k] / (sm[7] or 1), sm[k], sm[7], n.value, min(ms), " ".join("%.1f" % m for m in ms)))
