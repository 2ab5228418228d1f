#!/usr/bin/env python3
"""Diagnostic: per-section shader-clock breakdown of k3b_expand (needs libeg3d.so built with
EG3D_EXTRA_HIPFLAGS=-DEG3D_SECTION_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegraph3d_amd import api, host  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s = host.Synth(cfg)
ctx = api.Context(s.scene)
ctx.upload_seeds(s.seeds)
for _ in range(2):
    r = ctx.match_resident(0, s.n_seeds, device_only=True)
L = api.lib()
L.eg3d_probe_sections.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
sm, sl, n = (C.c_double * 16)(), (C.c_double * 16)(), C.c_uint32()
assert L.eg3d_probe_sections(ctx._h, sm, sl, C.byref(n)) == 0
names = ["candidates", "stepwalks", "sidewalks", "batchGN", "follow", "stepDLT", "stepGN", "whole", "commit", "init", "epc-pre", "newpoint"]
tot = sm[7] or 1
print("chains", n.value, "times", r["times"])
for k in range(12):
    print("%-11s sum %12.3e (%5.1f%%)   slowest chain %10.3e (%5.1f%%)" % (names[k], sm[k], 100 * sm[k] / tot, sl[k], 100 * sl[k] / (sl[7] or 1)))
print("follow, all chains: sequential steps %d, look-ahead steps accepted %d, look-ahead rounds redone %d" % (sm[12], sm[13], sm[14]))
print("follow, slowest chain (%d points): sequential steps %d, look-ahead steps accepted %d, redone %d" % (sl[15], sl[12], sl[13], sl[14]))
print("mean ticks per chain %.3e ; slowest %.3e" % (sm[7] / max(1, n.value), sl[7]))

L.eg3d_probe_hyp_sections.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
hs_, hl_, cnt = (C.c_double * 6)(), (C.c_double * 6)(), (C.c_uint32 * 5)()
assert L.eg3d_probe_hyp_sections(ctx._h, hs_, hl_, cnt) == 0
print("hypotheses n/tri/d1/d2/compat:", list(cnt))
print("follow lists: total n1 %d n2 %d ; longest n1 %d n2 %d ; hypotheses with >=32 points %d" % (hs_[0], hs_[1], hl_[0], hl_[1], hs_[2]))
