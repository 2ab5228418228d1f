#!/usr/bin/env python3
"""GPU box: how well do contiguous seed shards balance the per-rank time of a C4 step, for N = 2 / 4 / 8 ranks, under
different split weights? A rank's share of a step is EMULATED on the one GPU: the shard's
seeds run alone (eg3d_match_resident on that range, device-only) and the whole-call device time (and K3b's) is
recorded. Weights: "k" = sum of track lengths (round 2), "k2" = sum of k^2, "tasks_k" = stage-A tasks of the seed
x k (what k_chain_cost uses per chain; needs a stage-A pre-pass: eg3d_candidates_run here).
-> gpurun_out/shard_balance.json.  usage: shard_balance.py [batches=2] [per_rank=4096]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgegraph3d_amd import api, host  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
s = host.Synth(4)
off = s.seeds_np()[0].astype(np.int64)
k = np.diff(off)
ctx = api.Context(s.scene)
ctx.upload_seeds(s.seeds)


def cuts_for(weight, b0, b1, world):
    c = np.concatenate([[0], np.cumsum(weight[b0:b1])])
    cuts = [b0]
    for r in range(1, world):
        t = c[-1] * r // world
        cuts.append(min(max(int(np.searchsorted(c, t, side="left")) + b0, cuts[-1]), b1))
    cuts.append(b1)
    return cuts


rows = []
for world in (2, 4, 8):
    batch = per_rank * world
    for bi in range(n_batches):
        b0 = bi * batch
        b1 = min(s.n_seeds, b0 + batch)
        if b1 - b0 < batch:
            break
        # stage-A tasks per seed of this batch (pre-pass), in pieces to bound the host copy
        tasks = np.zeros(s.n_seeds, np.int64)
        for p0 in range(b0, b1, 1024):
            c = ctx.candidates(s.seeds, p0, min(b1, p0 + 1024))
            sv_seed = np.repeat(np.arange(p0, min(b1, p0 + 1024)), k[p0:min(b1, p0 + 1024)])
            np.add.at(tasks, sv_seed[c["task_sv"]], 1)
        weights = {"k": k, "k2": k * k, "tasks_k": tasks * k}
        for name, w in weights.items():
            cuts = cuts_for(w.astype(np.int64), b0, b1, world)
            ms_total, ms_expand, pts = [], [], []
            for r in range(world):
                best = None
                for rep in range(2):  # second run: buffers sized
                    res = ctx.match_resident(cuts[r], cuts[r + 1], device_only=True)
                    t = res["times"]
                    if best is None or t["ms_total"] < best[0]:
                        best = (t["ms_total"], t["ms_expand"], res["n_points"])
                ms_total.append(best[0]); ms_expand.append(best[1]); pts.append(best[2])
            row = {"ranks": world, "batch": [b0, b1], "weight": name, "seeds_per_rank": [cuts[r + 1] - cuts[r] for r in range(world)],
                   "ms_total_per_rank": [round(x, 1) for x in ms_total], "ms_expand_per_rank": [round(x, 1) for x in ms_expand],
                   "max_over_mean_total": round(max(ms_total) / (sum(ms_total) / world), 3),
                   "max_over_mean_expand": round(max(ms_expand) / (sum(ms_expand) / world), 3), "points_per_rank": pts}
            print(row, flush=True)
            rows.append(row)
            os.makedirs("gpurun_out", exist_ok=True)
            json.dump(rows, open("gpurun_out/shard_balance.json", "w"), indent=1)
summary = {}
for world in (2, 4, 8):
    for name in ("k", "k2", "tasks_k"):
        v = [r["max_over_mean_total"] for r in rows if r["ranks"] == world and r["weight"] == name]
        if v:
            summary["N=%d %s" % (world, name)] = {"mean_max_over_mean": round(float(np.mean(v)), 3), "worst": max(v)}
print(summary)
json.dump({"rows": rows, "summary": summary}, open("gpurun_out/shard_balance.json", "w"), indent=1)
