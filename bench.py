#!/usr/bin/env python3
"""bench.py — throughput of the refpoint -> epipolar match -> triangulate hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (eg3d_match_resident: K1..K4) over one batch of synthetic
seeds whose scene and tracks are already resident in HBM. N=1 runs BASELINE.json configs[1]
(C2: 8 views / 2000 seeds / ~5k polyline segments per view). For N>1 (launched by
torch.distributed.run, one rank per GPU) every rank owns its own 2000-seed shard of the same
scene (weak scaling) and the step ends with an RCCL all-gather of the edge-point cloud over xGMI.
Rank 0 prints ONE JSON line. value = whole-job edge-points per second.

Besides the contract fields the line carries `roofline` (dominant kernel, HBM bound, algorithmic
bytes of SURVEY 8(d) / that kernel's HIP-event duration) and `cpu_baseline` (the CPU oracle,
1 thread, timed on this box on the same workload; N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
SEEDS_PER_GPU = 2000
STAGES = [("k1_seed_candidates", "ms_candidates"), ("k2_epipolar_hits", "ms_epipolar"),
          ("k3a_hypotheses", "ms_hypotheses"), ("k3s_select", "ms_select"), ("k3b_expand", "ms_expand"),
          ("k4_emit", "ms_emit")]


class _DevArr:
    """Expose a raw HBM pointer to torch through __cuda_array_interface__ (no copy)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=2, help="synthetic config index (2 = C2, BASELINE configs[1])")
    ap.add_argument("--seeds-per-gpu", type=int, default=0, help="override the per-GPU seed count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the RCCL all-gather of the cloud even with one rank (exercises the N>1 code path)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="independent steps kept in flight per GPU, each on its own context/HIP stream driven by its "
                         "own host thread (1 = strictly one step at a time)")
    ap.add_argument("--path", choices=["refpoints", "sets"], default="refpoints",
                    help="refpoints = pipeline 3 (the headline path); sets = the pipelines 1-2 extractor (SURVEY N1) on "
                         "one synthetic polyline set per 3-D curve (single GPU only)")
    ap.add_argument("--cpu-runs", type=int, default=5)
    ap.add_argument("--cpu-seeds", type=int, default=0,
                    help="bound the CPU baseline to the first K seeds of the workload (0 = all); its rate is "
                         "edge-points of that sample / its time")
    args = ap.parse_args()

    # Several steps are kept in flight on separate HIP streams (plus the gather stream and RCCL's):
    # with the runtime's default of 4 hardware queues two of them can share a queue and serialise
    # (measured: the all-gather of one step then waits for another step's whole expand kernel).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    from edgegraph3d_amd import api, host

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, args.gpus))
    dist = None
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if api.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = host.default_config(args.config)
    per_gpu = args.seeds_per_gpu or cfg.n_seeds
    if args.config == 2 and not args.seeds_per_gpu:
        per_gpu = SEEDS_PER_GPU
    cfg.n_seeds = per_gpu * world  # same scene on every rank; rank r owns seeds [r*per_gpu, (r+1)*per_gpu)
    synth = host.Synth(cfg)
    b, e = rank * per_gpu, (rank + 1) * per_gpu
    sets = synth.polyline_sets() if args.path == "sets" else None
    if sets is not None and world > 1:
        raise SystemExit("bench.py --path sets is a single-GPU measurement")
    inflight = max(1, args.inflight)

    # One context (own HIP stream, own work buffers) + one host thread per step in flight: the
    # tail of one step's chain expansion (a few long chains on an otherwise idle GPU) overlaps
    # with the next step's candidate search and hypothesis evaluation. ctypes releases the GIL
    # during the library call, so the threads really run concurrently.
    import queue
    import threading

    class Worker(threading.Thread):
        def __init__(self, parent=None):
            super().__init__(daemon=True)
            if parent is None:
                self.ctx = api.Context(synth.scene, local_rank)
                self.ctx.upload_seeds(synth.seeds)   # inputs resident in HBM before the timed region
            else:
                self.ctx = parent.ctx.clone()        # shares the resident scene and seeds (eg3d_clone)
            self.todo, self.done = queue.Queue(), queue.Queue()
            self.start()

        def run(self):
            while self.todo.get() is not None:
                try:
                    if sets is not None:
                        self.done.put(self.ctx.match_polyline_sets(sets[0], sets[1], sets[2], device_only=True))
                    else:
                        self.done.put(self.ctx.match_resident(b, e, device_only=True))
                except Exception as ex:  # surfaced by the main thread
                    self.done.put(ex)

    workers = [Worker()]
    workers += [Worker(workers[0]) for _ in range(inflight - 1)]

    gather = None
    if dist is not None:
        from edgegraph3d_amd.distributed import CloudGather
        gather = CloudGather(dist, world, dev)
        # the gather's small pack/unpack kernels share the GPU with a saturating expand kernel of
        # another step: give them a high-priority stream so they are dispatched as slots free up
        gstream = torch.cuda.Stream(device=dev, priority=-1)

    def allgather_cloud(ctx):
        """RCCL all-gather of the variable-length edge-point cloud straight from the context's HBM
        buffers (edgegraph3d_amd/distributed.py): counts, one padded all_gather_into_tensor, then
        compaction into one ordered cloud on every rank."""
        d = ctx.last_device_output()
        if not d.complete:
            raise RuntimeError("bench: output spans several chunks; shrink the per-GPU batch")
        np_, no_ = int(d.n_points), int(d.n_obs)
        local = {}
        for name, ptr, per, n in (("X", d.X, 12, np_), ("obs_off", d.obs_off, 4, np_), ("key", d.key, 16, np_),
                                  ("obs_view", d.obs_view, 4, no_), ("obs_pl", d.obs_pl, 4, no_),
                                  ("obs_seg", d.obs_seg, 4, no_), ("obs_xy", d.obs_xy, 8, no_)):
            local[name] = torch.as_tensor(_DevArr(ptr, max(n, 1) * per), device=dev)
        recv, counts, layout = gather.allgather(local, np_, no_)
        cloud = gather.unpack(recv, counts, layout)
        return cloud["n_points"]

    def run_steps(n, pool):
        """n steps, at most len(pool) in flight; results (and the collectives, which must be issued
        in the same order on every rank) are handled in step order by this thread."""
        out, submitted = [], 0
        for w in pool[:n]:
            w.todo.put(1)
            submitted += 1
        for i in range(n):
            w = pool[i % len(pool)]
            r = w.done.get()
            if isinstance(r, Exception):
                raise r
            total = r["n_points"]
            if gather is not None:
                with torch.cuda.stream(gstream):
                    total = allgather_cloud(w.ctx)
                gather.wait_pack()  # the cloud has left the context's buffers; its all-gather may still be in flight
            if submitted < n:
                w.todo.put(1)
                submitted += 1
            out.append((r, total))
        return out

    run_steps(max(args.warmup, inflight), workers)  # every context sizes its buffers before the timed region
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = run_steps(args.steps, workers)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = {k: [r["times"][k] for r, _ in results] for _, k in STAGES}
    last, total_points = results[-1]
    bytes_alg = last["times"]["bytes_algorithmic"]
    # the same steps strictly one at a time (untimed side measurement, reported beside the value)
    single = None
    if inflight > 1 and rank == 0 and world == 1:
        torch.cuda.synchronize()
        ts = time.perf_counter()
        ns = max(3, min(10, args.steps))
        run_steps(ns, workers[:1])
        torch.cuda.synchronize()
        single = (time.perf_counter() - ts) / ns
    ctx = workers[0].ctx

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_points * args.steps / elapsed
        avg = {k: (sum(v) / len(v) if v else 0.0) for k, v in stage_ms.items()}
        dom_name, dom_key = max(STAGES, key=lambda s: avg[s[1]])
        dom_ms = avg[dom_key]
        achieved = (bytes_alg / (dom_ms * 1e-3)) / 1e9 if dom_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and args.config == 2 and not args.seeds_per_gpu:  # counters were collected on C2
            try:
                traffic = json.load(open(pmc)).get(dom_name, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "triangulated edge-points/sec", "value": value, "unit": "edge-points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "steps_in_flight": inflight,
            "config": {
                "workload": ("C%d synthetic: %d views / %d seeds per GPU / %.0f polyline segments per view"
                             % (args.config, synth.n_views, per_gpu, synth.total_segments / synth.n_views))
                if sets is None else
                ("C%d synthetic, pipelines 1-2 extractor: %d views / %d polyline sets (%d polylines) / %.0f segments "
                 "per view" % (args.config, synth.n_views, sets[0], len(sets[2]), synth.total_segments / synth.n_views)),
                "edge_points_per_step": int(total_points), "observations_per_step_rank0": int(last["n_obs"]),
                "tasks": int(last["n_tasks"]), "hypotheses": int(last["n_hypotheses"]), "chains": int(last["n_chains"]),
                "parallelism": "seed-shard x%d%s" % (world, " + RCCL all-gather of the cloud" if world > 1 else ""),
                "arithmetic": "2-D geometry f32, DLT+Gauss-Newton f64 (as the reference)",
            },
            "stage_ms": {n: round(avg[k], 4) for n, k in STAGES},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(bytes_alg), "kernel_ms": dom_ms},
        }
        if single is not None:
            line["one_step_at_a_time"] = {"ms_per_step": single * 1e3, "value": total_points / single}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import binding as ob   # cpu_baseline leg: the checker timed as the CPU port
            orc = ob.Oracle(synth.scene)
            if sets is not None:
                def cpu_run(lo, hi, nthreads):
                    return orc.match_polyline_sets(sets[0], sets[1], sets[2], lo, hi, nthreads)
                b, e = 0, sets[0]
                warm = min(e, 2)
            else:
                def cpu_run(lo, hi, nthreads):
                    return orc.match(synth.seeds, lo, hi, nthreads)
                warm = min(e, b + 200)
            cpu_run(b, warm, 1)  # warm-up
            ce = e if not args.cpu_seeds else min(e, b + args.cpu_seeds)
            secs, pts = [], 0
            for _ in range(max(1, args.cpu_runs)):
                r = cpu_run(b, ce, 1)
                secs.append(r["stats"]["seconds"])
                pts = r["n_points"]
            med = statistics.median(secs)
            ncores = os.cpu_count() or 1
            rall = cpu_run(b, ce, ncores)
            line["cpu_baseline"] = {
                "value": pts / med, "unit": "edge-points/s", "cores": 1, "kind": "port",
                "sample": "%s of the N=1 workload (%d %s, %d edge-points), oracle -O3, 1 thread, median of %d runs "
                          "(%.2f s each); scene/grid construction excluded"
                          % ("all" if ce == e else "first %d" % (ce - b), ce - b, "sets" if sets is not None else "seeds",
                             pts, len(secs), med),
                "all_cores": {"value": rall["n_points"] / rall["stats"]["seconds"], "cores": ncores},
                "same_point_count_as_gpu": (bool(pts == total_points) if ce == e else None),
                "oracle_algorithmic_bytes": int(r["stats"]["bytes_algorithmic"]),
            }
            line["speedup_vs_cpu_1thread"] = value / (pts / med)
            # "CPU-ref parity err" (the second half of BASELINE.json's metric): the step's output copied to the
            # host and compared with the oracle's output on the same seeds / sets — relative error of the 3-D
            # coordinates (north star: <= 1e-4) and exactness of every id / view list and of the order
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from parity_util import compare_edgepoints
            if sets is not None:
                gfull = workers[0].ctx.match_polyline_sets(sets[0], sets[1], sets[2], b, ce)
            else:
                gfull = workers[0].ctx.match_resident(b, ce)
            rep = compare_edgepoints(r, gfull, rel_tol=1e-4)
            line["parity"] = {"vs": "oracle (CPU restatement; parity unpinned, see DESIGN.md 3)", "points_compared": int(pts),
                              "max_rel_err_X": rep.get("max_rel_X"), "X_bit_exact": rep.get("bitexact_X"),
                              "ids_views_order_exact": bool(rep["ok"]), "obs_xy_bit_exact": rep.get("bitexact_xy"),
                              "tolerance": 1e-4}
        print(json.dumps(line), flush=True)
    for w in workers:
        w.todo.put(None)
        w.join(timeout=10)
        w.ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
