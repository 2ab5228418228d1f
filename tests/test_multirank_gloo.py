"""N>1 path on CPU: world_size-2 gloo. Each rank computes its contiguous, sum-of-track-length
balanced seed shard with the HOST SIMULATION of the device code (tests/hostsim: the kernels'
per-lane bodies of stage B compiled for the host; stage A from the oracle, as the hostsim takes
it), then the cloud is exchanged by HostCloudGather: gloo moves the raw arrays, the C functions of the
product (eg3d_host_gather_plan / eg3d_host_gather_place — the plan eg3d_allgather_edgepoints applies to
device buffers) place and rebase them. Every rank must reproduce the single-process oracle output
exactly; a rank without a result must make every rank return the same error.
(The RCCL exchange itself — include/eg3d_rccl.h — needs GPUs: tests/rccl_two_rank_check.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from edgegraph3d_amd import host
from edgegraph3d_amd.distributed import HostCloudGather, shard_range, shard_ranges_balanced


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _empty_cloud():
    return {"n_points": 0, "n_obs": 0, "X": np.zeros((0, 3), np.float32), "obs_off": np.zeros(1, np.uint64),
            "key": np.zeros((0, 4), np.uint32), "obs_view": np.zeros(0, np.int32), "obs_pl": np.zeros(0, np.uint32),
            "obs_seg": np.zeros(0, np.uint32), "obs_xy": np.zeros((0, 2), np.float32)}


def _worker(rank, world, port, cfg, q, empty_rank=-1, failing_rank=None):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import binding as ob
    import hostsim_binding as hs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = host.Synth(cfg)
    o = ob.Oracle(s.scene)
    if empty_rank < 0:
        b, e = shard_ranges_balanced(s.seeds_np()[0], 0, s.n_seeds, world)[rank]
    else:
        # one rank owns NO seeds (a step whose batch is smaller than the world, or an unlucky split): the others share
        # the range, balanced by track length
        others = shard_ranges_balanced(s.seeds_np()[0], 0, s.n_seeds, world - 1)
        b, e = (others[empty_rank][0],) * 2 if rank == empty_rank else others[rank - (1 if rank > empty_rank else 0)]
    # stage B = the device code run on the host; its keys carry the global seed index, as the GPU's do
    r = hs.match(s.scene, s.seeds, b, e, o.candidates_raw(s.seeds, b, e)) if e > b else _empty_cloud()
    g = HostCloudGather(dist, world, rank)
    cloud, rc = g.allgather(r)
    assert rc == 0
    # a rank whose match failed still takes part (local = None): EVERY rank gets EG3D_GATHER_ERR_INCOMPLETE
    bad = world - 1 if failing_rank is None else failing_rank
    none, rc_bad = g.allgather(None if rank == bad else r)
    assert none is None and rc_bad == -4
    q.put((rank, {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in cloud.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_balanced_shards_cover_and_balance_by_track_length():
    s = host.Synth(1)
    off = s.seeds_np()[0].astype(np.int64)
    for w in (1, 2, 3, 8, 200):
        r = shard_ranges_balanced(off, 0, s.n_seeds, w)
        assert len(r) == w and r[0][0] == 0 and r[-1][1] == s.n_seeds
        assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        if w <= 8:
            weights = [off[e] - off[b] for b, e in r]
            assert max(weights) - min(weights) <= 2 * int(np.diff(off).max())
    r = shard_ranges_balanced(off, 10, 50, 4)                      # a sub-range (one bench step's batch)
    assert r[0][0] == 10 and r[-1][1] == 50
    # a skewed set: one very long track — count-balanced shards would be 4x off, these are not
    skew = np.concatenate([[0], np.cumsum([300] + [3] * 99)])
    r = shard_ranges_balanced(skew, 0, 100, 2)
    assert r[0] == (0, 1) or (skew[r[0][1]] - skew[r[0][0]]) <= 303


def test_shard_ranges_cover_and_balance():
    for n, w in ((10, 3), (2000, 8), (5, 8), (0, 2)):
        r = shard_range(n, w)
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        sizes = [e - b for b, e in r]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,empty_rank,failing_rank", [(2, -1, None), (4, 1, 2)],
                         ids=["2 ranks", "4 ranks, rank 1 without seeds, rank 2 failing"])
def test_gloo_allgather_reproduces_single_process_output(world, empty_rank, failing_rank):
    from oracle import binding as ob
    cfg = 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cfg, q, empty_rank, failing_rank)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s = host.Synth(cfg)
    ref = ob.Oracle(s.scene).match(s.seeds, 0, s.n_seeds, 1)
    for rank in range(world):
        c = got[rank]
        assert c["n_points"] == ref["n_points"] and c["n_obs"] == ref["n_obs"]
        assert np.array_equal(c["X"].view(np.uint32), ref["X"].view(np.uint32))
        assert np.array_equal(c["key"].astype(np.uint32), ref["key"])
        assert np.array_equal(c["obs_off"].astype(np.uint32), ref["obs_off"])  # incl. the n_obs sentinel
        assert np.array_equal(c["obs_view"], ref["obs_view"])
        assert np.array_equal(c["obs_pl"].astype(np.uint32), ref["obs_pl"])
        assert np.array_equal(c["obs_seg"].astype(np.uint32), ref["obs_seg"])
        assert np.array_equal(c["obs_xy"].view(np.uint32), ref["obs_xy"].view(np.uint32))


def _worker_steps(rank, world, port, n_seeds, batch, q):
    """bench.py --gpus N on CPU ranks: the SAME StepPlan object Leg uses decides which seeds this rank takes in each
    step; the device code of stage B runs as its host simulation; HostCloudGather stands in for the RCCL exchange."""
    import hashlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import binding as ob
    import hostsim_binding as hs
    from edgegraph3d_amd.distributed import StepPlan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = host.default_config(4)        # BASELINE configs[3] shape: 200 views, ~20k segments per view
    cfg.n_seeds = n_seeds
    s = host.Synth(cfg)
    o = ob.Oracle(s.scene)
    plan = StepPlan(s.seeds_np()[0], s.n_seeds, batch, world, rank)
    g = HostCloudGather(dist, world, rank)
    digests = []
    for i in range(plan.n_pass_steps()):
        b, e = plan.pass_range(i)
        assert (b, e) == plan.step_range(i) or i >= plan.n_batches   # whole batches: the timed steps take the same ranges
        r = (hs.match(s.scene, s.seeds, b, e, o.candidates_raw(s.seeds, b, e), chain_cap=8192, pool_cap=1 << 18)
             if e > b else _empty_cloud())
        cloud, rc = g.allgather(r)
        assert rc == 0
        h = hashlib.sha256()
        for k in ("X", "obs_off", "key", "obs_view", "obs_pl", "obs_seg", "obs_xy"):
            dt = {"obs_off": np.uint64, "key": np.uint32, "obs_pl": np.uint32, "obs_seg": np.uint32}.get(k)
            a = np.ascontiguousarray(cloud[k] if dt is None else np.asarray(cloud[k]).astype(dt))
            h.update(a.view(np.uint8).tobytes())
        digests.append((int(cloud["n_points"]), int(cloud["n_obs"]), (b, e), h.hexdigest()))
    q.put((rank, digests))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_world_8_steps_of_a_c4_shaped_scene_with_the_bench_sharding():
    """8 ranks, a 200-view scene, 96 seeds in steps of 40 (two whole batches + a partial one: 5 seeds per rank and step,
    fewer in the last): every step's gathered cloud must be, on every rank, the oracle's cloud of that batch."""
    import hashlib
    from oracle import binding as ob
    from edgegraph3d_amd.distributed import StepPlan
    world, n_seeds, batch = 8, 96, 40
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_steps, args=(r, world, port, n_seeds, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=540) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = host.default_config(4)
    cfg.n_seeds = n_seeds
    s = host.Synth(cfg)
    o = ob.Oracle(s.scene)
    plan0 = StepPlan(s.seeds_np()[0], s.n_seeds, batch, world, 0)
    assert plan0.n_pass_steps() == 3 and plan0.n_batches == 2
    covered = []
    for i in range(plan0.n_pass_steps()):
        b0, b1 = plan0.batch_bounds(i, cyclic=False)
        ref = o.match(s.seeds, b0, b1, 8)
        h = hashlib.sha256()
        for k in ("X", "obs_off", "key", "obs_view", "obs_pl", "obs_seg", "obs_xy"):
            dt = {"obs_off": np.uint64, "key": np.uint32, "obs_pl": np.uint32, "obs_seg": np.uint32}.get(k)
            a = np.ascontiguousarray(ref[k] if dt is None else np.asarray(ref[k]).astype(dt))
            h.update(a.view(np.uint8).tobytes())
        ranges = [got[r][i][2] for r in range(world)]
        assert ranges[0][0] == b0 and ranges[-1][1] == b1 and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
        covered.append((b0, b1))
        for r in range(world):
            assert got[r][i][:2] == (ref["n_points"], ref["n_obs"]), (r, i)
            assert got[r][i][3] == h.hexdigest(), (r, i)
    assert covered[0][0] == 0 and covered[-1][1] == n_seeds
