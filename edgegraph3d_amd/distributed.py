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


class StepPlan:
    """Which seeds a rank takes in step i of the multi-GPU workload — the ONE statement of it: bench.py's Leg uses this
    object, and tests/test_multirank_gloo.py runs the same object on CPU ranks. A step = one batch of `batch` seeds
    (bench.py --gpus N: 4096 x N), split over the ranks into contiguous ranges balanced by the sum of track lengths;
    rank order = seed order, so the concatenation of the ranks' clouds is the batch's cloud."""

    def __init__(self, trk_off, n_total, batch, world, rank):
        self.trk_off, self.n_total, self.world, self.rank = trk_off, int(n_total), int(world), int(rank)
        self.batch = max(1, min(int(batch), self.n_total)) if self.n_total else 1
        self.n_batches = max(1, self.n_total // self.batch)   # whole batches: the timed steps cycle through them

    def batch_bounds(self, i, cyclic=True):
        b0 = ((i % self.n_batches) if cyclic else i) * self.batch
        return b0, min(b0 + self.batch, self.n_total)

    def step_range(self, i):
        """Timed step i: batch i (cyclic over the WHOLE batches), this rank's share."""
        b0, b1 = self.batch_bounds(i, True)
        return shard_ranges_balanced(self.trk_off, b0, b1, self.world)[self.rank]

    def pass_range(self, i):
        """Step i of ONE pass over all seeds: ceil(n / batch) steps, the last one partial."""
        b0, b1 = self.batch_bounds(i, False)
        return shard_ranges_balanced(self.trk_off, b0, b1, self.world)[self.rank]

    def n_pass_steps(self):
        return (self.n_total + self.batch - 1) // self.batch


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
        # pre-flight, part 1: what RCCL itself reports about the communicator must be what the launcher started
        self.G.eg3d_comm_query.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        n_, r_, d_ = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        rc = self.G.eg3d_comm_query(self.comm, C.byref(n_), C.byref(r_), C.byref(d_))
        self.comm_info = {"ncclCommCount": n_.value, "ncclCommUserRank": r_.value, "ncclCommCuDevice": d_.value}
        import sys as _sys
        print("RCCL-PREFLIGHT rank %d/%d: ncclCommCount=%d ncclCommUserRank=%d device=%d (rc %d)"
              % (rank, world, n_.value, r_.value, d_.value, rc), file=_sys.stderr, flush=True)
        if rc != 0 or n_.value != world or r_.value != rank or d_.value != device_index:
            self.close()
            raise RuntimeError("RCCL pre-flight: the communicator reports %s, expected %d ranks / rank %d / device %d"
                               % (self.comm_info, world, rank, device_index))
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
        self.G.eg3d_gather_set_mode.argtypes = [C.c_void_p, C.c_int]
        self.mode = "bcast" if os.environ.get("EG3D_GATHER_MODE", "")[:1] in ("b", "1") else "sendrecv"
        self.selftest = None
        if world > 1 and dist.get_backend() == "nccl":
            self._choose_mode(dist, device_index)

    # ---- pre-flight check of the exchange (multi-rank only) -------------------------------------------------
    def _selftest_once(self, device_index):
        """A tiny hand-made cloud per rank (rank r: 2 + r points; odd ranks' points carry two observations each, even
        ranks' none — zero-size arrays on some pairs) through eg3d_allgather_edgepoints; True when this rank holds exactly
        the concatenation every rank can predict."""
        import numpy as np
        import torch as _t
        C, D = self.C, self.D
        dev = _t.device("cuda", device_index)

        def cloud(r):
            n = 2 + r
            k = 2 if r % 2 else 0
            X = (np.arange(3 * n, dtype=np.float32) + 1000.0 * r).reshape(n, 3)
            key = (np.arange(4 * n, dtype=np.int64) + 7 * r).astype(np.uint32).reshape(n, 4)
            off = (np.arange(n + 1, dtype=np.uint64) * k)
            m = n * k
            view = (np.arange(m, dtype=np.int32) + r)
            pl = (np.arange(m, dtype=np.int64) * 3 + r).astype(np.uint32)
            seg = (np.arange(m, dtype=np.int64) * 5 + r).astype(np.uint32)
            xy = (np.arange(2 * m, dtype=np.float32) * 0.5 + r).reshape(m, 2)
            return X, off, key, view, pl, seg, xy

        mine = cloud(self.rank)
        keep = [_t.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev) if a.size else
                _t.zeros(16, dtype=_t.uint8, device=dev) for a in mine]
        loc = D.DeviceEdgePoints()
        loc.n_points, loc.n_obs, loc.complete = len(mine[0]), len(mine[3]), 1
        loc.X, loc.obs_off, loc.key, loc.obs_view, loc.obs_pl, loc.obs_seg, loc.obs_xy = [t.data_ptr() for t in keep]
        out = D.DeviceEdgePoints()
        _t.cuda.synchronize(dev)
        rc = self.G.eg3d_allgather_edgepoints(self.g, self.comm, self.world, self.rank, self.stream, C.byref(loc), C.byref(out),
                                              self.rank_points, self.rank_obs)
        if rc != 0:
            return rc, False
        parts = [cloud(r) for r in range(self.world)]
        want_X = np.concatenate([p[0] for p in parts])
        want_key = np.concatenate([p[2] for p in parts])
        obase, want_off = 0, [np.zeros(1, np.uint64)]
        for p in parts:
            want_off.append(p[1][1:] + np.uint64(obase))
            obase += len(p[3])
        want_off = np.concatenate(want_off)
        want = [want_X, want_off, want_key] + [np.concatenate([p[i] for p in parts]) for i in (3, 4, 5, 6)]
        if int(out.n_points) != len(want_X) or int(out.n_obs) != obase:
            return 0, False
        try:
            hip = C.CDLL("libamdhip64.so.7")
        except OSError:
            hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        ptrs = [out.X, out.obs_off, out.key, out.obs_view, out.obs_pl, out.obs_seg, out.obs_xy]
        for w, ptr in zip(want, ptrs):
            w = np.ascontiguousarray(w)
            got = np.empty_like(w)
            if w.nbytes and hip.hipMemcpy(got.ctypes.data, C.c_void_p(ptr), w.nbytes, 2) != 0:
                return 0, False
            if not np.array_equal(got.view(np.uint8), w.view(np.uint8)):
                return 0, False
        return 0, True

    def _choose_mode(self, dist, device_index):
        """The grouped send/recv exchange is the default; if the pre-flight cloud does not come out right on EVERY rank
        (an RCCL build on which it misbehaves), all ranks switch to the broadcast fallback together and test that."""
        import torch as _t
        dev = _t.device("cuda", device_index)
        tried = []
        for mode in ([self.mode] if self.mode == "bcast" else ["sendrecv", "bcast"]):
            self.G.eg3d_gather_set_mode(self.g, 1 if mode == "bcast" else 0)
            rc, ok = self._selftest_once(device_index)
            # [0] = ranks whose data came out wrong, [1] = ranks whose call FAILED (rc != 0: with EG3D_GATHER_ERR_FATAL the
            # library has already aborted the communicator on that rank — it must not be used again, by anyone)
            t = _t.tensor([0 if ok or rc != 0 else 1, 1 if rc != 0 else 0], dtype=_t.int32, device=dev)
            dist.all_reduce(t)
            bad_data, failed = int(t[0].item()), int(t[1].item())
            tried.append((mode, bad_data == 0 and failed == 0))
            if failed:
                # a failed call is not a reason to try the other mode on the same communicator: give it up on every rank
                self.selftest = tried
                if rc == -6:
                    self.abandon_comm()
                self.close()
                raise RuntimeError("eg3d_allgather_edgepoints failed (rc %d on this rank, %d rank(s) failed) in its pre-flight "
                                   "check, mode %s: the communicator is abandoned" % (rc, failed, mode))
            if tried[-1][1]:
                self.mode = mode
                self.selftest = tried
                return
        self.selftest = tried
        self.close()
        raise RuntimeError("eg3d_allgather_edgepoints failed its pre-flight check in every exchange mode: %s" % tried)

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
