#!/usr/bin/env python3
"""Diagnostic (timing build, EG3D_LIB=edgegraph3d_amd/variants/libeg3d_timing.so): where the waves of the K3a engine
(eg3d_k3a_engine.h) spend their shader clocks. usage: k3a_stats.py <cfg> [n_seeds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegraph3d_amd import api, host
cfg = int(sys.argv[1]); ns = int(sys.argv[2]) if len(sys.argv) > 2 else 0
s = host.Synth(cfg)
ctx = api.Context(s.scene); ctx.upload_seeds(s.seeds)
n = ns or s.n_seeds
L = api.lib(); buf = (C.c_ulonglong * 128)()
ctx.match_resident(0, n, device_only=True)
L.eg3d_probe_gn(buf, 1)
ctx.match_resident(0, n, device_only=True)
L.eg3d_probe_gn(buf, 1)
b = list(buf)
for k, name in ((0, "k3a_orient"), (1, "k3a_follow_spec")):
    adv, srv, con, its, req, work, passes = b[113 + 7 * k: 120 + 7 * k]
    tot = max(1, adv + srv + con)
    print("%-16s wave-clocks: advance %.1f%%  serve %.1f%%  consume %.1f%%  (total %.3e)" % (name, 100 * adv / tot, 100 * srv / tot, 100 * con / tot, tot))
    print("                 iterations %d  requests %d (%.1f of 64 slots per iteration)  working lanes per iteration %.1f  clocks per iteration %.0f (serve %.0f)"
          % (its, req, req / max(1, its), work / max(1, its), tot / max(1, its), srv / max(1, its)))
