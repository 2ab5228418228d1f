"""Multi-GPU plumbing of the path: seed sharding and the all-gather of the edge-point cloud.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU for
tests). Seeds are independent units, so rank r owns the contiguous range
shard_ranges_balanced(...)[r] (balanced by the sum of track lengths; shard_range = by count)
and the concatenation of the per-rank outputs in rank order IS the single-process output. The
only exchange step is the variable-length all-gather of the cloud: counts first (16 B/rank),
then ONE padded all_gather_into_tensor of the packed SoA
    [X(12 B/pt) | obs_off(4) | key(16) | obs_view(4/obs) | obs_pl(4) | obs_seg(4) | obs_xy(8)]
— a single large message per rank so every xGMI link carries traffic at once.
"""
import torch

POINT_FIELDS = (("X", 12), ("obs_off", 4), ("key", 16))
OBS_FIELDS = (("obs_view", 4), ("obs_pl", 4), ("obs_seg", 4), ("obs_xy", 8))
_DTYPES = {"X": torch.float32, "obs_off": torch.int32, "key": torch.int32, "obs_view": torch.int32,
           "obs_pl": torch.int32, "obs_seg": torch.int32, "obs_xy": torch.float32}


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
            raise RuntimeError("eg3d_gather_create failed")
        self.stream = C.c_void_p(stream_ptr) if stream_ptr else None
        self.rank_points = (C.c_uint64 * world)()
        self.rank_obs = (C.c_uint64 * world)()

    def allgather(self, local_dev):
        """local_dev = Context.last_device_output(). Returns (DeviceEdgePoints of the whole cloud, rc);
        every rank gets the same rc (see include/eg3d_rccl.h)."""
        C = self.C
        out = self.D.DeviceEdgePoints()
        rc = self.G.eg3d_allgather_edgepoints(self.g, self.comm, self.world, self.rank, self.stream,
                                              C.byref(local_dev), C.byref(out), self.rank_points, self.rank_obs)
        return out, rc

    def close(self):
        if self.g:
            self.G.eg3d_gather_destroy(self.g)
            self.g = None
        if self.comm:
            self.G.eg3d_comm_destroy(self.comm)
            self.comm = self.C.c_void_p()


class CloudGather:
    """Reusable staging buffers + the two collectives. `local` maps field name -> 1-D uint8
    tensor (raw bytes, on `device`) holding this rank's n_points / n_obs elements."""

    def __init__(self, dist, world, device):
        self.dist, self.world, self.device = dist, world, device
        self.cap = 0
        self.send = self.recv = None
        self.pack_done = None  # event recorded once `local` has been copied into the send buffer

    def allgather(self, local, n_points, n_obs):
        dist, world, dev = self.dist, self.world, self.device
        cnt = torch.tensor([n_points, n_obs], dtype=torch.int64, device=dev)
        allc = torch.empty(2 * world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, cnt)
        counts = allc.view(world, 2).cpu()
        mp, mo = int(counts[:, 0].max()), int(counts[:, 1].max())
        nbytes = mp * sum(s for _, s in POINT_FIELDS) + mo * sum(s for _, s in OBS_FIELDS)
        nbytes = max(nbytes, 16)
        if self.cap < nbytes:
            self.cap = int(nbytes * 1.25) + 256
            self.send = torch.empty(self.cap, dtype=torch.uint8, device=dev)
            self.recv = torch.empty(self.cap * world, dtype=torch.uint8, device=dev)
        send = self.send[:nbytes]
        o = 0
        for name, per in POINT_FIELDS:
            if n_points:
                send[o:o + n_points * per].copy_(local[name][:n_points * per])
            o += mp * per
        for name, per in OBS_FIELDS:
            if n_obs:
                send[o:o + n_obs * per].copy_(local[name][:n_obs * per])
            o += mo * per
        if dev.type == "cuda":
            # the producer may overwrite `local` (the context's output buffers) once this has fired;
            # the collective itself then overlaps with the producer's next step
            self.pack_done = torch.cuda.Event()
            self.pack_done.record()
        recv = self.recv[:nbytes * world]
        dist.all_gather_into_tensor(recv, send)
        return recv, counts, (mp, mo, nbytes)

    def wait_pack(self):
        """Block the host until the last allgather() no longer reads its `local` buffers."""
        if self.pack_done is not None:
            self.pack_done.synchronize()
            self.pack_done = None

    def unpack(self, recv, counts, layout):
        """Compact the padded per-rank blocks into one cloud (rank order = seed order); obs_off is
        rebased so it indexes the concatenated observation arrays."""
        mp, mo, nbytes = layout
        parts = {n: [] for n, _ in POINT_FIELDS + OBS_FIELDS}
        obs_base = 0
        for r in range(self.world):
            np_, no_ = int(counts[r, 0]), int(counts[r, 1])
            blk = recv[r * nbytes:(r + 1) * nbytes]
            o = 0
            for name, per in POINT_FIELDS:
                t = blk[o:o + np_ * per].view(_DTYPES[name])
                if name == "obs_off":
                    t = t + obs_base
                parts[name].append(t)
                o += mp * per
            for name, per in OBS_FIELDS:
                parts[name].append(blk[o:o + no_ * per].view(_DTYPES[name]))
                o += mo * per
            obs_base += no_
        out = {n: torch.cat(v) for n, v in parts.items()}
        # n_obs sentinel, as the C ABI (eg3d_edgepoints.obs_off[n_points]) and the RCCL path return it
        out["obs_off"] = torch.cat([out["obs_off"], torch.tensor([obs_base], dtype=out["obs_off"].dtype,
                                                                  device=out["obs_off"].device)])
        out["X"] = out["X"].view(-1, 3)
        out["key"] = out["key"].view(-1, 4)
        out["obs_xy"] = out["obs_xy"].view(-1, 2)
        out["n_points"] = int(counts[:, 0].sum())
        out["n_obs"] = int(counts[:, 1].sum())
        return out
