#!/usr/bin/env python3
"""Two-rank check of the C-ABI exchange step (include/eg3d_rccl.h: eg3d_allgather_edgepoints) for a
node with >= 2 GPUs — the single-GPU boxes of the round cannot run it, tests/test_gpu_parity.py
skips it there. No torch: the communicator comes from eg3d_comm_unique_id / eg3d_comm_init (ncclCommInitRank), the
unique id travels through a file.

    python tests/rccl_two_rank_check.py            # spawns ranks 0 and 1 (GPUs 0 and 1), prints RCCL-2RANK-OK
    python tests/rccl_two_rank_check.py --world 4  # more ranks if the node has them

Cases, on the small synthetic scene:
  1. unequal shards (sum-of-track-length balanced split of a skewed seed range): the gathered
     cloud on EVERY rank equals the single-process result bit for bit, obs_off rebased;
  2. a rank with ZERO seeds;
  3. a rank whose local result is marked incomplete: every rank returns EG3D_GATHER_ERR_INCOMPLETE
     (-4) from the same call instead of blocking in the payload collective;
  4. hand-made clouds: one rank with points but NO observations, one rank with nothing, one with both
     (zero-size arrays must be skipped symmetrically on both sides of every pair).
Cases 1, 2 and 4 run in both exchange modes (grouped send/recv, and the broadcast fallback) and with the
transfers cut into 4 KiB pieces as well as whole (the path clouds of more than 1 GiB per array take).
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def worker(rank, world, idfile):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from edgegraph3d_amd import _cdefs as D
    from edgegraph3d_amd import api, host
    from edgegraph3d_amd.distributed import shard_ranges_balanced
    pkg = os.path.dirname(os.path.abspath(api.__file__))
    G = C.CDLL(os.path.join(pkg, "libeg3d_rccl.so"))
    hip = C.CDLL("libamdhip64.so")
    uid = UniqueId()
    if rank == 0:
        assert G.eg3d_comm_unique_id(C.byref(uid)) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.rename(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
        C.memmove(C.byref(uid), open(idfile, "rb").read(), 128)
    comm = C.c_void_p()
    G.eg3d_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    assert G.eg3d_comm_init(C.byref(uid), world, rank, rank, C.byref(comm)) == 0   # hipSetDevice(rank) + ncclCommInitRank
    G.eg3d_gather_create.restype = C.c_void_p
    G.eg3d_gather_create.argtypes = [C.c_int]
    G.eg3d_allgather_edgepoints.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                            C.POINTER(D.DeviceEdgePoints), C.POINTER(D.DeviceEdgePoints),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    G.eg3d_gather_destroy.argtypes = [C.c_void_p]
    g = G.eg3d_gather_create(rank)
    assert g
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def fetch(ptr, n, dtype):
        a = np.empty(n, dtype)
        if n:
            assert hip.hipMemcpy(a.ctypes.data, C.cast(ptr, C.c_void_p), a.nbytes, 2) == 0
        return a

    s = host.Synth(1)
    ctx = api.Context(s.scene, rank)
    ctx.upload_seeds(s.seeds)
    whole = ctx.match_resident(0, s.n_seeds)            # the single-process answer, computed on every rank
    off = s.seeds_np()[0]
    rp, ro = (C.c_uint64 * world)(), (C.c_uint64 * world)()

    def gather_and_check(ranges, label):
        b, e = ranges[rank]
        ctx.match_resident(b, e, device_only=True)
        local = ctx.last_device_output()
        out = D.DeviceEdgePoints()
        rc = G.eg3d_allgather_edgepoints(g, comm, world, rank, None, C.byref(local), C.byref(out), rp, ro)
        assert rc == 0, (label, rc)
        lo, hi = ranges[0][0], ranges[-1][1]
        sel = (whole["key"][:, 0] >= lo) & (whole["key"][:, 0] < hi)
        n, m = int(out.n_points), int(out.n_obs)
        assert n == int(sel.sum()), (label, n, int(sel.sum()))
        first = int(np.argmax(sel)) if n else 0
        o0 = int(whole["obs_off"][first]) if n else 0
        assert m == int(whole["obs_off"][first + n]) - o0
        assert np.array_equal(fetch(out.X, 3 * n, np.uint32), whole["X"][first:first + n].view(np.uint32).ravel())
        assert np.array_equal(fetch(out.key, 4 * n, np.uint32), whole["key"][first:first + n].ravel())
        assert np.array_equal(fetch(out.obs_off, n + 1, np.uint64), whole["obs_off"][first:first + n + 1] - o0)
        assert np.array_equal(fetch(out.obs_view, m, np.int32), whole["obs_view"][o0:o0 + m])
        assert np.array_equal(fetch(out.obs_pl, m, np.uint32), whole["obs_pl"][o0:o0 + m])
        assert np.array_equal(fetch(out.obs_seg, m, np.uint32), whole["obs_seg"][o0:o0 + m])
        assert np.array_equal(fetch(out.obs_xy, 2 * m, np.uint32), whole["obs_xy"][o0:o0 + m].view(np.uint32).ravel())
        assert sum(rp) == n and sum(ro) == m
        print("rank %d: %s ok (%d points, per rank %s)" % (rank, label, n, list(rp)), flush=True)

    G.eg3d_gather_set_mode.argtypes = [C.c_void_p, C.c_int]
    G.eg3d_gather_set_chunk_bytes.argtypes = [C.c_void_p, C.c_uint64]
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]

    def to_dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), max(16, a.nbytes)) == 0
        if a.nbytes:
            assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p

    def synthetic_clouds(label):
        """rank r holds: r % 3 == 0 -> 7 points without observations; == 1 -> nothing; == 2 -> 5 points x 2 observations."""
        kind = rank % 3
        npts, nobs = (7, 0) if kind == 0 else (0, 0) if kind == 1 else (5, 10)
        rng = np.random.default_rng(100 + rank)
        X = rng.random((npts, 3), dtype=np.float32)
        key = rng.integers(0, 1 << 30, (npts, 4), dtype=np.uint32)
        ooff = (np.arange(npts + 1, dtype=np.uint64) * (2 if kind == 2 else 0))
        view = rng.integers(0, 8, nobs, dtype=np.int32)
        pl = rng.integers(0, 1000, nobs, dtype=np.uint32)
        seg = rng.integers(0, 50, nobs, dtype=np.uint32)
        xy = rng.random((nobs, 2), dtype=np.float32)
        loc = D.DeviceEdgePoints()
        loc.n_points, loc.n_obs, loc.complete = npts, nobs, 1
        loc.X, loc.obs_off, loc.key = to_dev(X), to_dev(ooff), to_dev(key)
        loc.obs_view, loc.obs_pl, loc.obs_seg, loc.obs_xy = to_dev(view), to_dev(pl), to_dev(seg), to_dev(xy)
        out = D.DeviceEdgePoints()
        rc = G.eg3d_allgather_edgepoints(g, comm, world, rank, None, C.byref(loc), C.byref(out), rp, ro)
        assert rc == 0, (label, rc)
        # expected: every rank's arrays regenerated from its seed
        eX, ekey, eoff, eview, exy = [], [], [np.zeros(1, np.uint64)], [], []
        obase = 0
        for r in range(world):
            k = r % 3
            n_p, n_o = (7, 0) if k == 0 else (0, 0) if k == 1 else (5, 10)
            g2 = np.random.default_rng(100 + r)
            eX.append(g2.random((n_p, 3), dtype=np.float32))
            ekey.append(g2.integers(0, 1 << 30, (n_p, 4), dtype=np.uint32))
            eoff.append(obase + (np.arange(1, n_p + 1, dtype=np.uint64) * (2 if k == 2 else 0)))
            eview.append(g2.integers(0, 8, n_o, dtype=np.int32))
            g2.integers(0, 1000, n_o, dtype=np.uint32)
            g2.integers(0, 50, n_o, dtype=np.uint32)
            exy.append(g2.random((n_o, 2), dtype=np.float32))
            obase += n_o
        n, m = int(out.n_points), int(out.n_obs)
        assert n == sum(len(a) for a in eX) and m == obase, (label, n, m)
        assert np.array_equal(fetch(out.X, 3 * n, np.float32), np.concatenate(eX).ravel())
        assert np.array_equal(fetch(out.key, 4 * n, np.uint32), np.concatenate(ekey).ravel())
        assert np.array_equal(fetch(out.obs_off, n + 1, np.uint64), np.concatenate(eoff))
        assert np.array_equal(fetch(out.obs_view, m, np.int32), np.concatenate(eview))
        assert np.array_equal(fetch(out.obs_xy, 2 * m, np.float32), np.concatenate(exy).ravel())
        print("rank %d: %s ok (%d points, %d observations)" % (rank, label, n, m), flush=True)

    for mode, mname in ((0, "send/recv"), (1, "bcast")):
        for chunk in (1 << 30, 4096):
            assert G.eg3d_gather_set_mode(g, mode) == 0 and G.eg3d_gather_set_chunk_bytes(g, chunk) == 0
            tag = " [%s, pieces of %d B]" % (mname, chunk)
            # 1. unequal shards
            gather_and_check(shard_ranges_balanced(off, 0, s.n_seeds, world), "balanced shards" + tag)
            uneven = [(0, 5)] + [(5 + (s.n_seeds - 5) * (r - 1) // (world - 1), 5 + (s.n_seeds - 5) * r // (world - 1))
                                 for r in range(1, world)]
            gather_and_check(uneven, "uneven shards" + tag)
            # 2. a rank with zero seeds
            zero = [(0, 0)] + [(s.n_seeds * (r - 1) // (world - 1), s.n_seeds * r // (world - 1)) for r in range(1, world)]
            gather_and_check(zero, "empty rank 0" + tag)
            # 4. points without observations / nothing / both
            synthetic_clouds("hand-made clouds" + tag)
    assert G.eg3d_gather_set_mode(g, 0) == 0 and G.eg3d_gather_set_chunk_bytes(g, 1 << 30) == 0
    # 3. an incomplete local result on the LAST rank only: every rank gets the same error code
    b, e = shard_ranges_balanced(off, 0, s.n_seeds, world)[rank]
    ctx.match_resident(b, e, device_only=True)
    local = ctx.last_device_output()
    if rank == world - 1:
        local.complete = 0
    out = D.DeviceEdgePoints()
    rc = G.eg3d_allgather_edgepoints(g, comm, world, rank, None, C.byref(local), C.byref(out), rp, ro)
    assert rc == -4, rc
    print("rank %d: incomplete rank reported to every rank (rc -4)" % rank, flush=True)
    G.eg3d_gather_destroy(g)
    G.eg3d_comm_destroy.argtypes = [C.c_void_p]
    G.eg3d_comm_destroy(comm)
    ctx.close()


def main():
    if "--rank" in sys.argv:
        i = sys.argv.index("--rank")
        worker(int(sys.argv[i + 1]), int(sys.argv[i + 2]), sys.argv[i + 3])
        return
    world = int(sys.argv[sys.argv.index("--world") + 1]) if "--world" in sys.argv else 2
    with tempfile.TemporaryDirectory() as d:
        idfile = os.path.join(d, "nccl_id")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), str(world), idfile])
                 for r in range(world)]
        rcs = [p.wait(timeout=600) for p in procs]
    if any(rcs):
        raise SystemExit("rccl two-rank check failed: %s" % rcs)
    print("RCCL-2RANK-OK")


if __name__ == "__main__":
    main()
