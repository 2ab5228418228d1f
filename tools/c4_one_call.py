#!/usr/bin/env python3
"""GPU box: BASELINE config 4's WHOLE cloud from ONE call: eg3d_match_resident(0, 100000,
device_only=1) on C4 -> 56 375 432 edge-points / 4 109 039 077 observations (more than 2^32: the cloud's
observation offsets are 64-bit), compared batch by batch (13 x 8192 seeds, computed on a second context) bit for
bit on the host, then eg3d_concat_edgepoints of the 13 batch clouds against the same per-batch data.
-> gpurun_out/c4_one_call.json.  usage: c4_one_call.py [n_seeds=100000] [batch=8192]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # device storage for the batch clouds of the concat leg; initialised BEFORE libeg3d.so loads its HIP runtime
torch.cuda.init()
from edgegraph3d_amd import _cdefs as D, api, host  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
s = host.Synth(4)
n_seeds = min(n_seeds, s.n_seeds)
FIELDS = ("X", "key", "obs_off", "obs_view", "obs_pl", "obs_seg", "obs_xy")


def same(a, b):
    return all(np.array_equal(np.ascontiguousarray(a[k]).view(np.uint8), np.ascontiguousarray(b[k]).view(np.uint8)) for k in FIELDS)


rep = {"workload": "C4 (BASELINE configs[3]): 200 views, seeds [0, %d) in ONE eg3d_match_resident call (device_only)" % n_seeds}
A = api.Context(s.scene)
A.upload_seeds(s.seeds)
t0 = time.time()
r = A.match_resident(0, n_seeds, device_only=True)
rep["one_call_seconds"] = round(time.time() - t0, 1)
rep["points"], rep["observations"], rep["flags"] = int(r["n_points"]), int(r["n_obs"]), int(r["flags"])
rep["observations_exceed_2^32"] = rep["observations"] > 0xffffffff
print(rep, flush=True)
try:
    want = json.load(open(os.path.join(ROOT, "profiles", "r02_c4_full_parity.json")))
    if n_seeds == 100000:
        rep["equals_r02_full_parity_totals"] = (rep["points"] == want["total_points"] and rep["observations"] == want["total_observations"])
except OSError:
    pass
devA = A.last_device_output()
assert devA.complete
# ---- batch by batch on a second context: bit-exact against the corresponding slice of the one-call cloud
B = A.clone()
hip = C.CDLL("libamdhip64.so.7")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
keep, parts_host_digest, p0, rows = [], [], 0, []
PER = {"X": 12, "obs_off": 8, "key": 16, "obs_view": 4, "obs_pl": 4, "obs_seg": 4, "obs_xy": 8}
for lo in range(0, n_seeds, batch):
    hi = min(n_seeds, lo + batch)
    rb = B.match_resident(lo, hi, device_only=True)
    devB = B.last_device_output()
    nb = int(devB.n_points)
    gotB = B.fetch_device_points(devB, 0, nb)
    gotA = A.fetch_device_points(devA, p0, p0 + nb)
    ok = same(gotA, gotB)
    rows.append({"seeds": [lo, hi], "points": nb, "observations": int(devB.n_obs), "slice_of_one_call_bit_exact": bool(ok)})
    print(rows[-1], flush=True)
    assert ok
    # keep the batch cloud resident (torch tensors) for the concat leg
    t = {}
    for k in FIELDS:
        n_el = nb if k in ("X", "obs_off", "key") else int(devB.n_obs)
        t[k] = torch.empty(max(1, n_el * PER[k]), dtype=torch.uint8, device="cuda")
        if n_el:
            assert hip.hipMemcpy(C.c_void_p(t[k].data_ptr()), C.cast(getattr(devB, k), C.c_void_p), n_el * PER[k], 3) == 0
    keep.append((t, nb, int(devB.n_obs)))
    parts_host_digest.append({k: (int(np.ascontiguousarray(gotB[k]).view(np.uint8).astype(np.uint64).sum()), gotB[k].shape) for k in FIELDS})
    p0 += nb
    del gotA, gotB
rep["batches"] = rows
rep["all_slices_bit_exact"] = all(x["slice_of_one_call_bit_exact"] for x in rows) and p0 == rep["points"]
A.close()
B.close()
torch.cuda.synchronize()
# ---- eg3d_concat_edgepoints of the batch clouds == the one-call cloud (checked through the per-batch digests)
G = C.CDLL(os.path.join(ROOT, "edgegraph3d_amd", "libeg3d_rccl.so"))
G.eg3d_gather_create.restype = C.c_void_p
G.eg3d_gather_create.argtypes = [C.c_int]
G.eg3d_gather_destroy.argtypes = [C.c_void_p]
G.eg3d_concat_edgepoints.argtypes = [C.c_void_p, C.c_int, C.POINTER(D.DeviceEdgePoints), C.c_void_p, C.POINTER(D.DeviceEdgePoints)]
g = G.eg3d_gather_create(0)
parts = (D.DeviceEdgePoints * len(keep))()
for i, (t, nb, no) in enumerate(keep):
    parts[i].n_points, parts[i].n_obs, parts[i].complete = nb, no, 1
    for k in FIELDS:
        setattr(parts[i], k, t[k].data_ptr())
cat = D.DeviceEdgePoints()
t0 = time.time()
rc = G.eg3d_concat_edgepoints(g, len(keep), parts, None, C.byref(cat))
rep["concat_rc"], rep["concat_seconds"] = rc, round(time.time() - t0, 2)
assert rc == 0, rc
rep["concat_points"], rep["concat_observations"] = int(cat.n_points), int(cat.n_obs)
X = api.Context(s.scene)  # only for its fetch helper
p0, ok_all = 0, True
for i, (t, nb, no) in enumerate(keep):
    got = X.fetch_device_points(cat, p0, p0 + nb)
    dg = {k: (int(np.ascontiguousarray(got[k]).view(np.uint8).astype(np.uint64).sum()), got[k].shape) for k in FIELDS}
    ok_all = ok_all and dg == parts_host_digest[i] and got["n_obs"] == no
    p0 += nb
rep["concat_equals_batches"] = bool(ok_all and rep["concat_points"] == rep["points"] and rep["concat_observations"] == rep["observations"])
G.eg3d_gather_destroy(g)
print({k: v for k, v in rep.items() if k != "batches"}, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rep, open("gpurun_out/c4_one_call.json", "w"), indent=1)
assert rep["all_slices_bit_exact"] and rep["concat_equals_batches"]
