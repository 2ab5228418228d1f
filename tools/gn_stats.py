#!/usr/bin/env python3
"""Diagnostic (timing build): Gauss-Newton iteration statistics of k3b_expand. usage: gn_stats.py <cfg> [n_seeds]"""
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
print("cfg", cfg, "requests", b[64], "accepted", b[65], "rows/request %.1f" % (b[66] / max(1, b[64])), "rounds", b[69], "req/round %.2f" % (b[64] / max(1, b[69])))
print("row-iterations useful", b[68], " lane-iterations held", b[67], " utilisation %.3f" % (b[68] / max(1, b[67])))
print("requests by iterations:", {i: b[i] for i in range(32) if b[i]})
print("rounds by iterations:  ", {i: b[32 + i] for i in range(32) if b[32 + i]})
print("long requests", b[102], "long rounds", b[103], "by iterations:", {i: b[70 + i] for i in range(32) if b[70 + i]})
names = ["rows", "product stores+barrier", "pass-1 sums", "gsum+inverse", "pass-2 products", "pass-2 sums", "update", "round total"]
tot = b[111] or 1
print("ticks inside gn_round:", {names[i]: "%.1f%%" % (100.0 * b[104 + i] / tot) for i in range(7)}, "round total %.3e" % b[111], "coop_gn_groups total %.3e (set-up share %.1f%%)" % (b[112], 100.0 * (b[112] - b[111]) / max(1, b[112])))
