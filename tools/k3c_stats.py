#!/usr/bin/env python3
"""Diagnostic (timing build, EG3D_LIB=edgegraph3d_amd/variants/libeg3d_timing.so, EG3D_K3B_ENGINE=1): where the waves of
the lane-per-chain engine (eg3d_k3c_engine.h) spend their shader clocks. usage: k3c_stats.py <cfg> [n_seeds]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("EG3D_K3B_ENGINE", "1")
from edgegraph3d_amd import api, host
cfg = int(sys.argv[1]); ns = int(sys.argv[2]) if len(sys.argv) > 2 else 0
s = host.Synth(cfg)
ctx = api.Context(s.scene); ctx.upload_seeds(s.seeds)
n = ns or s.n_seeds
L = api.lib(); buf = (C.c_ulonglong * 128)(); gn = (C.c_ulonglong * 128)()
ctx.match_resident(0, n, device_only=True)
L.eg3d_probe_k3c(buf, 1); (hasattr(L, "eg3d_probe_gn") and L.eg3d_probe_gn(gn, 1))
r = ctx.match_resident(0, n, device_only=True)
L.eg3d_probe_k3c(buf, 1); (hasattr(L, "eg3d_probe_gn") and L.eg3d_probe_gn(gn, 1))
b = list(buf); g = list(gn)
names = ["fetch", "advance", "pack", "solve drain", "candidate drain"]
tot = max(1, sum(b[:5]))
print("k3c_engine cfg %d: k3b_expand stage %.2f ms; wave clocks %.3e: " % (cfg, r["times"]["ms_expand"], tot)
      + "  ".join("%s %.1f%%" % (names[i], 100.0 * b[i] / tot) for i in range(5)))
its = max(1, b[8])
print("  iterations %d (%.0f clocks each); per iteration: lanes with a chain %.1f, blocked on solves %.1f, on candidates %.1f; "
      "solves %.1f in %.2f windows, candidate items %.1f" % (b[8], tot / its, b[9] / its, b[13] / its, b[14] / its, b[10] / its, b[12] / its, b[11] / its))
rounds = max(1, g[69])
print("  solver: %d requests in %d rounds (%.2f per round), rows per request %.2f, row fill %.2f of 64, "
      "clocks per round %.0f; share of solve drain inside rounds %.1f%%"
      % (g[64], g[69], g[64] / rounds, g[66] / max(1, g[64]), g[68] / max(1.0, g[67] / 64.0) , g[111] / rounds, 100.0 * g[111] / max(1, b[3])))

STATES = ["VIEW_NEXT", "EPC_POST", "EPC_LOOP", "CAND_POST", "CAND_DONE", "VISIT", "VISIT_ATTACH", "ATTACH_BEGIN", "CENTRAL_DONE",
          "ATTACH_SIDES", "SIDES_DONE", "ATTACH_CHECK", "FOLLOW_SIDE", "FOLLOW_STEP", "STEP_CAND", "STEP_TRI", "STEP_TRI_DONE",
          "FB_NEXT", "FB_TRI_DONE", "FB_ADD", "FB_ADD_DONE", "STEP_OK", "FOLLOW_END", "ATTACH_RET", "FINISH"]
adv = max(1, b[1])
print("  blocks of the advance (share of its wall clocks; lanes inside; clocks per entry):")
for i, nm in enumerate(STATES):
    w, l, n = b[32 + 3 * i: 35 + 3 * i]
    if n:
        print("    %-14s %5.1f%%  lanes %.2f  entries %9d  clocks/entry %8.0f" % (nm, 100.0 * w / adv, l / max(1, w), n, w / n))
t = g[104:113]
tt = max(1, g[111])
print("  inside solver rounds: rows %.1f%%  barrier %.1f%%  sums-1 %.1f%%  inverse %.1f%%  products-2 %.1f%%  sums-2 %.1f%%  update %.1f%%"
      % tuple(100.0 * x / tt for x in t[:7]))
