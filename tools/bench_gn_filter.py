#!/usr/bin/env python3
"""Config 5 (BASELINE configs[4]): gaussNewtonFiltering on ~1 M synthetic points — k5_gn_filter
throughput (kernel HIP-event time, H2D/D2H excluded) with its HBM roofline, the oracle timed
beside it, and a bit-exact parity check on the whole batch.

  python tools/bench_gn_filter.py [n_points=1000000] [runs=5]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegraph3d_amd import api, host  # noqa: E402

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(_pos[0]) if len(_pos) > 0 else 1000000
runs = int(_pos[1]) if len(_pos) > 1 else 5
s = host.Synth(5)   # 16-view rig: k really spans 3..10
X, off, view, xy = s.points(n)
ctx = api.Context(s.scene)
ms = []
for _ in range(runs + 1):
    Xo, inl, m = ctx.gn_filter(X, off, view, xy, 2.25)
    ms.append(m)
ms = ms[1:]
k_ms = float(np.median(ms))
n_obs = int(off[-1])
# algorithmic bytes: X in/out (12+12), inlier (1), obs_off (4), per observation view id + xy (12)
alg = n * (12 + 12 + 1 + 4) + n_obs * 12
line = {"workload": "C5 synthetic: %d points, %d observations (k~U[3,10], mean %.2f), 16-view rig" % (n, n_obs, n_obs / n),
        "kernel": "k5_gn_filter", "kernel_ms": k_ms, "points_per_s": n / (k_ms * 1e-3), "inlier_frac": float(inl.mean()),
        "roofline": {"bound": "hbm", "algorithmic_bytes": alg, "achieved_GBps": alg / (k_ms * 1e-3) / 1e9, "peak_GBps": 8000.0,
                     "frac": alg / (k_ms * 1e-3) / 1e9 / 8000.0}}
if "--no-cpu" not in sys.argv:
    from oracle import binding as ob
    o = ob.Oracle(s.scene)
    m = min(n, 200000)
    t = time.time()
    Xr, ir = o.gn_filter(X[:m], off[:m + 1], view[:off[m]], xy[:off[m]], 2.25, nthreads=1)
    dt = time.time() - t
    line["cpu_baseline"] = {"value": m / dt, "unit": "points/s", "cores": 1, "kind": "port", "sample": "first %d points" % m}
    line["parity_bitexact_on_sample"] = bool(np.array_equal(inl[:m], ir) and np.array_equal(Xo[:m].view(np.uint32), Xr.view(np.uint32)))
    line["speedup_vs_cpu_1thread"] = line["points_per_s"] / (m / dt)
    # parity on ALL points (oracle on every host thread: not a timing)
    Xa, ia = o.gn_filter(X, off, view, xy, 2.25, nthreads=os.cpu_count() or 1)
    line["parity_bitexact_all_points"] = bool(np.array_equal(inl, ia) and np.array_equal(Xo.view(np.uint32), Xa.view(np.uint32)))
print(json.dumps(line))
ctx.close()
