"""Single-rank check of include/eg3d_rccl.h (run as a script in its OWN process by
tests/test_gpu_parity.py: /opt/rocm's librccl must not share a process with the HIP/HSA runtime
bundled in the torch wheel, which other test modules import)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgegraph3d_amd import api, host  # noqa: E402


def main():
    """include/eg3d_rccl.h: the C-ABI all-gather of the cloud (counts, one padded ncclAllGather of the
    packed SoA from the context's HBM buffers, device compaction). With one rank the gathered cloud
    must equal the rank's own output; the N-rank ordering logic is covered on CPU by
    tests/test_multirank_gloo.py."""
    import ctypes as C
    import os
    from edgegraph3d_amd import _cdefs as D
    pkg = os.path.dirname(os.path.abspath(api.__file__))
    G = C.CDLL(os.path.join(pkg, "libeg3d_rccl.so"))
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    uid = (C.c_ubyte * 128)()
    assert G.eg3d_comm_unique_id(uid) == 0
    comm = C.c_void_p()
    G.eg3d_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    assert G.eg3d_comm_init(uid, 1, 0, 0, C.byref(comm)) == 0
    # default: the small scene; EG3D_GATHER_CHECK="<config> <seed_begin> <seed_end>" for a large one (ad hoc, e.g.
    # "4 0 8192": a whole multi-chunk C4 step, a 7 GB cloud through pack / all-gather / unpack)
    spec = os.environ.get("EG3D_GATHER_CHECK", "1").split()
    s = host.Synth(int(spec[0]))
    b, e = (int(spec[1]), int(spec[2])) if len(spec) == 3 else (0, s.n_seeds)
    ctx = api.Context(s.scene)
    ctx.upload_seeds(s.seeds)
    want = ctx.match_resident(b, e)
    ctx.match_resident(b, e, device_only=True)
    local = ctx.last_device_output()
    G.eg3d_gather_create.restype = C.c_void_p
    G.eg3d_allgather_edgepoints.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                            C.POINTER(D.DeviceEdgePoints), C.POINTER(D.DeviceEdgePoints),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    G.eg3d_gather_destroy.argtypes = [C.c_void_p]
    g = G.eg3d_gather_create(0)
    out = D.DeviceEdgePoints()
    rp, ro = (C.c_uint64 * 1)(), (C.c_uint64 * 1)()
    rc = G.eg3d_allgather_edgepoints(g, comm, 1, 0, None, C.byref(local), C.byref(out), rp, ro)
    assert rc == 0 and out.n_points == want["n_points"] and out.n_obs == want["n_obs"] and rp[0] == want["n_points"]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def fetch(ptr, n, dtype):
        a = np.empty(n, dtype)
        assert hip.hipMemcpy(a.ctypes.data, C.cast(ptr, C.c_void_p), a.nbytes, 2) == 0
        return a
    n, m = int(out.n_points), int(out.n_obs)
    assert np.array_equal(fetch(out.X, 3 * n, np.uint32), want["X"].view(np.uint32).ravel())
    assert np.array_equal(fetch(out.obs_off, n + 1, np.uint64), want["obs_off"])
    assert np.array_equal(fetch(out.key, 4 * n, np.uint32), want["key"].ravel())
    assert np.array_equal(fetch(out.obs_view, m, np.int32), want["obs_view"])
    assert np.array_equal(fetch(out.obs_pl, m, np.uint32), want["obs_pl"])
    assert np.array_equal(fetch(out.obs_seg, m, np.uint32), want["obs_seg"])
    assert np.array_equal(fetch(out.obs_xy, 2 * m, np.uint32), want["obs_xy"].view(np.uint32).ravel())
    # ---- the N-rank shape of the exchange on ONE GPU: eg3d_concat_edgepoints packs several resident clouds into
    # padded per-part slots and runs the same compaction kernel (rebased offsets, r > 0 blocks). Parts = unequal
    # seed ranges computed on cloned contexts, one of them EMPTY; the result must be the single-call cloud.
    def check_equal(o, w, label):
        nn, mm = int(o.n_points), int(o.n_obs)
        assert nn == w["n_points"] and mm == w["n_obs"], (label, nn, w["n_points"])
        assert np.array_equal(fetch(o.X, 3 * nn, np.uint32), w["X"].view(np.uint32).ravel()), label
        assert np.array_equal(fetch(o.obs_off, nn + 1, np.uint64), w["obs_off"]), label
        assert np.array_equal(fetch(o.key, 4 * nn, np.uint32), w["key"].ravel()), label
        assert np.array_equal(fetch(o.obs_view, mm, np.int32), w["obs_view"]), label
        assert np.array_equal(fetch(o.obs_pl, mm, np.uint32), w["obs_pl"]), label
        assert np.array_equal(fetch(o.obs_seg, mm, np.uint32), w["obs_seg"]), label
        assert np.array_equal(fetch(o.obs_xy, 2 * mm, np.uint32), w["obs_xy"].view(np.uint32).ravel()), label

    G.eg3d_concat_edgepoints.argtypes = [C.c_void_p, C.c_int, C.POINTER(D.DeviceEdgePoints), C.c_void_p,
                                         C.POINTER(D.DeviceEdgePoints)]
    span = e - b
    for cuts in ((0.1, 0.1, 0.55), (0.5,), (0.0, 0.33, 0.34, 0.9)):
        edges = [b] + [b + int(span * c) for c in cuts] + [e]
        ctxs = [ctx] + [ctx.clone() for _ in range(len(edges) - 2)]
        parts = (D.DeviceEdgePoints * (len(edges) - 1))()
        for i, cx in enumerate(ctxs):
            cx.match_resident(edges[i], edges[i + 1], device_only=True)
            parts[i] = cx.last_device_output()
        cat = D.DeviceEdgePoints()
        rc = G.eg3d_concat_edgepoints(g, len(ctxs), parts, None, C.byref(cat))
        assert rc == 0, rc
        check_equal(cat, want, "concat %s" % (cuts,))
        for cx in ctxs[1:]:
            cx.close()
    # a part that views the gather's OWN result buffers (the `out` of an earlier call on g) is refused, not read
    # after it has been freed or overwritten (include/eg3d_rccl.h)
    ctx.match_resident(b, e, device_only=True)
    alias = (D.DeviceEdgePoints * 2)()
    alias[0] = cat
    alias[1] = ctx.last_device_output()
    assert G.eg3d_concat_edgepoints(g, 2, alias, None, C.byref(D.DeviceEdgePoints())) == -1
    G.eg3d_gather_destroy(g)
    G.eg3d_comm_destroy.argtypes = [C.c_void_p]
    G.eg3d_comm_destroy(comm)
    ctx.close()
    # the pre-flight check bench.py --gpus N runs before trusting the exchange (edgegraph3d_amd/distributed.py): its
    # hand-made cloud must come back exactly, in both exchange modes (one rank: the placement / sentinel path)
    from edgegraph3d_amd.distributed import RcclCloudGather
    rg = RcclCloudGather(None, 1, 0, 0)
    for mode in (0, 1):
        assert rg.G.eg3d_gather_set_mode(rg.g, mode) == 0
        assert rg._selftest_once(0) == (0, True), "pre-flight cloud differs (mode %d)" % mode
    rg.close()


if __name__ == "__main__":
    main()
    print("RCCL-GATHER-OK")
