"""Multi-GPU plumbing of the path: seed sharding and the all-gather of the edge-point cloud.

One process per GPU (torch.distributed for rendezvous; RCCL over xGMI for the exchange; "gloo" on CPU
for tests). Seeds are independent units, so rank r owns the contiguous range
shard_ranges_balanced(...)[r] (balanced by the sum of track lengths; shard_range = by count)
and the concatenation of the per-rank outputs in rank order IS the single-process output. The
only exchange step is the variable-length all-gather of the cloud (include/eg3d_rccl.h): counts first
(24 B/rank), then the seven arrays of every rank's cloud
    X(12 B/pt) | obs_off(8) | key(16) | obs_view(4/obs) | obs_pl(4) | obs_seg(4) | obs_xy(8)
travel straight to their final position in every receiver's result arrays (no packing or padding), and
the observation offsets of ranks > 0 are rebased. RcclCloudGather = the C-ABI entry point on GPUs;
HostCloudGather = the same plan on host arrays around any torch.distributed backend.
"""
import torch

FIELDS = (("X", 12, True), ("obs_off", 8, True), ("key", 16, True), ("obs_view", 4, False), ("obs_pl", 4, False),
          ("obs_seg", 4, False), ("obs_xy", 8, False))  # name, bytes per element, per point (else per observation)


def shard_range(n_seeds, world):
    """Contiguous, count-balanced seed ranges [(begin, end)] per rank."""
    base, rem = divmod(n_seeds, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e))
        b = e
    return out


def shard_ranges_balanced(trk_off, begin, end, world):
    """Contiguous seed ranges of [begin, end) per rank, balanced by the sum of the track lengths k
    (the loop being split, plg_matching_from_refpoints.cpp:83-104, costs ~k candidate searches and
    ~k start views per seed) instead of by seed count. trk_off = the seeds' CSR offsets, so the
    weight of a range is one subtraction. Every rank computes the same split from the same array."""
    import numpy as np
    off = np.asarray(trk_off, dtype=np.int64)
    lo, hi = int(off[begin]), int(off[end])
    cuts = [begin]
    for r in range(1, world):
        target = lo + (hi - lo) * r // world
        c = int(np.searchsorted(off[begin:end + 1], target, side="left")) + begin
        cuts.append(min(max(c, cuts[-1]), end))
    cuts.append(end)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class RcclCloudGather:
    """The C-ABI exchange step (include/eg3d_rccl.h: eg3d_allgather_edgepoints in libeg3d_rccl.so) on an
    RCCL communicator created by the same library (eg3d_comm_init: hipSetDevice + ncclCommInitRank); the
    unique id travels through the caller's existing torch.distributed group (any backend)."""

    def __init__(self, dist, world, rank, device_index, stream_ptr=None):
        import ctypes as C
        import os
        import torch as _t
        from . import _cdefs as D
        self.C, self.D = C, D
        self.world, self.rank = world, rank
        self.g, self.comm = None, None
        pkg = os.path.dirname(os.path.abspath(__file__))
        self.G = C.CDLL(os.path.join(pkg, "libeg3d_rccl.so"))
        self.G.eg3d_comm_unique_id.argtypes = [C.c_void_p]
        self.G.eg3d_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        self.G.eg3d_comm_destroy.argtypes = [C.c_void_p]
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            rc = self.G.eg3d_comm_unique_id(uid)
            if rc != 0:
                raise RuntimeError("eg3d_comm_unique_id failed (%d)" % rc)
        if world > 1:
            dev = _t.device("cuda", device_index) if dist.get_backend() == "nccl" else _t.device("cpu")
            t = _t.tensor(list(bytes(uid)), dtype=_t.uint8, device=dev)
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().numpy().tobytes())
            C.memmove(uid, raw, 128)
        self.comm = C.c_void_p()
        rc = self.G.eg3d_comm_init(uid, world, rank, device_index, C.byref(self.comm))
        if rc != 0:
            raise RuntimeError("eg3d_comm_init (hipSetDevice + ncclCommInitRank) failed (%d) on rank %d" % (rc, rank))
        self.G.eg3d_gather_create.restype = C.c_void_p
        self.G.eg3d_gather_create.argtypes = [C.c_int]
        self.G.eg3d_gather_destroy.argtypes = [C.c_void_p]
        self.G.eg3d_allgather_edgepoints.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                     C.POINTER(D.DeviceEdgePoints), C.POINTER(D.DeviceEdgePoints),
                                                     C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        self.g = self.G.eg3d_gather_create(device_index)
        if not self.g:
            self.close()  # do not leak the communicator
            raise RuntimeError("eg3d_gather_create failed")
        self.stream = C.c_void_p(stream_ptr) if stream_ptr else None
        self.rank_points = (C.c_uint64 * world)()
        self.rank_obs = (C.c_uint64 * world)()

    def allgather(self, local_dev):
        """local_dev = Context.last_device_output(), or None when this rank has no usable result (its match
        failed): it still takes part, and the status word makes EVERY rank return EG3D_GATHER_ERR_INCOMPLETE (-4)
        instead of leaving the others inside the collective. Returns (DeviceEdgePoints of the whole cloud, rc);
        every rank gets the same rc (see include/eg3d_rccl.h)."""
        C = self.C
        out = self.D.DeviceEdgePoints()
        rc = self.G.eg3d_allgather_edgepoints(self.g, self.comm, self.world, self.rank, self.stream,
                                              C.byref(local_dev) if local_dev is not None else None, C.byref(out),
                                              self.rank_points, self.rank_obs)
        if rc == -6:
            self.abandon_comm()
        return out, rc

    def close(self):
        if getattr(self, "g", None):
            self.G.eg3d_gather_destroy(self.g)
            self.g = None
        if getattr(self, "comm", None):
            self.G.eg3d_comm_destroy(self.comm)
            self.comm = self.C.c_void_p()

    def abandon_comm(self):
        """After EG3D_GATHER_ERR_FATAL (-6) the library has aborted the communicator: forget it."""
        self.comm = self.C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostCloudGather:
    """The exchange on HOST arrays over any torch.distributed backend (gloo on CPU): the transport is one
    broadcast per (rank, array) of the raw bytes; the plan (bases, totals, status) and the placement + rebasing
    of every rank's cloud are the C functions eg3d_host_gather_plan / eg3d_host_gather_place of libeg3d_host.so —
    the same arithmetic eg3d_allgather_edgepoints applies to device buffers."""

    def __init__(self, dist, world, rank):
        from . import host as _host
        self.dist, self.world, self.rank = dist, world, rank
        self.H = _host.lib()

    def allgather(self, local):
        """local = a cloud dict as the C ABI returns it (obs_off with its sentinel), or None for "this rank has
        no usable result". Returns (whole cloud dict or None, rc), the same rc on every rank."""
        import ctypes as C
        import numpy as np
        from . import _cdefs as D
        dist, world, rank = self.dist, self.world, self.rank
        mine = [int(local["n_points"]), int(local["n_obs"]), 0] if local is not None else [0, 0, 1]
        allc = torch.empty(3 * world, dtype=torch.int64)
        dist.all_gather_into_tensor(allc, torch.tensor(mine, dtype=torch.int64))
        counts = np.ascontiguousarray(allc.numpy().astype(np.uint64))
        pbase = np.zeros(world, np.uint64)
        obase = np.zeros(world, np.uint64)
        tp, to = C.c_uint64(), C.c_uint64()
        self.H.eg3d_host_gather_plan.argtypes = [C.c_int, D.u64p, D.u64p, D.u64p, D.u64p, D.u64p]
        rc = self.H.eg3d_host_gather_plan(world, D.np_ptr(counts, C.c_uint64), D.np_ptr(pbase, C.c_uint64),
                                          D.np_ptr(obase, C.c_uint64), C.byref(tp), C.byref(to))
        if rc != 0:
            return None, rc
        tp, to = int(tp.value), int(to.value)
        whole = {"X": np.zeros((tp, 3), np.float32), "obs_off": np.zeros(tp + 1, np.uint64),
                 "key": np.zeros((tp, 4), np.uint32), "obs_view": np.zeros(to, np.int32), "obs_pl": np.zeros(to, np.uint32),
                 "obs_seg": np.zeros(to, np.uint32), "obs_xy": np.zeros((to, 2), np.float32)}
        whole_c = D.EdgePointsArrays(whole)
        whole_c.c.n_points, whole_c.c.n_obs = tp, to
        self.H.eg3d_host_gather_place.argtypes = [C.POINTER(D.EdgePoints), C.c_uint64, C.c_uint64, C.POINTER(D.EdgePoints)]
        dtypes = {"X": np.float32, "obs_off": np.uint64, "key": np.uint32, "obs_view": np.int32, "obs_pl": np.uint32,
                  "obs_seg": np.uint32, "obs_xy": np.float32}
        for r in range(world):
            np_r, no_r = int(counts[3 * r]), int(counts[3 * r + 1])
            part = {}
            for name, per, per_point in FIELDS:
                n_el = np_r if per_point else no_r
                nbytes = n_el * per
                if r == rank:
                    a = np.ascontiguousarray(local[name] if name != "obs_off" else local[name][:np_r], dtypes[name])
                    buf = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()) if nbytes else torch.zeros(0, dtype=torch.uint8)
                else:
                    buf = torch.empty(nbytes, dtype=torch.uint8)
                if nbytes:
                    dist.broadcast(buf, src=r)
                part[name] = buf.numpy().view(dtypes[name]) if nbytes else np.zeros(0, dtypes[name])
            part["obs_off"] = np.concatenate([part["obs_off"], np.array([no_r], np.uint64)])  # EdgePointsArrays wants the sentinel
            part_c = D.EdgePointsArrays(part)
            part_c.c.n_points, part_c.c.n_obs = np_r, no_r
            rc = self.H.eg3d_host_gather_place(C.byref(part_c.c), int(pbase[r]), int(obase[r]), C.byref(whole_c.c))
            if rc != 0:
                return None, rc
        out = dict(whole_c.a)
        out["obs_off"][tp] = to
        out["X"] = out["X"].reshape(-1, 3)
        out["key"] = out["key"].reshape(-1, 4)
        out["obs_xy"] = out["obs_xy"].reshape(-1, 2)
        out["n_points"], out["n_obs"] = tp, to
        return out, 0
