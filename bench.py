#!/usr/bin/env python3
"""bench.py — throughput of the refpoint -> epipolar match -> triangulate hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload auto|c2|c3|c4]

A "step" = one pass of the hot path (eg3d_match_resident: K1..K4) over one batch of synthetic
seeds whose scene and tracks are already resident in HBM.

  N = 1 (default workload c3): C3' = the dtu006-shaped configuration the north-star target is
      quoted on (25 views / 6268 seeds / ~15k polyline segments per view; BASELINE configs[2] with
      synthetic polylines — the reference's input.json for the real images is missing). One step
      = all 6268 seeds.
  N > 1 (default workload c4, launched by torch.distributed.run, one rank per GPU): BASELINE
      configs[3], 200 views / 100 000 seeds / ~20k segments per view, STRONG scaling: one step =
      one batch of 8192 seeds (steps cycle through the 12 whole batches of the 100k seeds), split
      over the ranks into contiguous ranges balanced by the sum of track lengths; every rank holds
      the whole scene; the step ends with the RCCL all-gather of the edge-point cloud through the
      C ABI (eg3d_allgather_edgepoints, include/eg3d_rccl.h). `--gpus 1 --workload c4` measures
      the same workload on one GPU (the base of the scaling curve).

Rank 0 prints ONE JSON line. value = whole-job edge-points per second with `steps_in_flight`
independent steps overlapped per GPU (K steps are still exactly K passes); the same steps strictly
one at a time are reported beside it (value_one_step_at_a_time / ms_per_step_one_at_a_time), and
`end_to_end` adds the D2H copy of the cloud. `roofline` = dominant kernel against the HBM peak
(algorithmic bytes of SURVEY 8(d) / that kernel's HIP-event time per step; traffic = PMC bytes from
the committed rocprofv3 passes of this workload, with provenance). `cpu_baseline` = the CPU oracle,
1 thread, on this box (N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import queue
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
C4_BATCH = 8192        # seeds per step of the c4 workload on one GPU (12 whole batches in 100 000 seeds)
C4_RANK_SHARE = 4096   # ... and per RANK on several: a step's batch = 4096 x ranks seeds. A chain of this workload runs
                       # for up to ~0.45 s on its own wavefront, so a rank needs a few thousand seeds per step to keep its
                       # GPU full behind such tails (measured on one GPU, 4 steps in flight: 1024 / 2048 / 4096 / 8192 seeds
                       # per step -> 0.98 / 1.70 / 1.94 / 1.98 M edge-points/s). The job (100 000 seeds) is the same at every N.
STAGES = [("k1_seed_candidates", "ms_candidates"), ("k2_epipolar_hits", "ms_epipolar"),
          ("k3a_hypotheses", "ms_hypotheses"), ("k3s_select", "ms_select"), ("k3b_expand", "ms_expand"),
          ("k4_emit", "ms_emit")]
WORKLOADS = {"c2": 2, "c3": 3, "c4": 4}


class _RealEdges:
    """BASELINE configs[2] from the REAL edge images: the 25 dtu006 edge maps (tests/golden/dtu006_edges) turned
    into polyline graphs by the N2 builder (eg3d_plg_build_from_mask), with synthetic look-at cameras at the 25
    listed camera centres and 6268 seeds sampled on the polylines (tests/real_scene.py). The reference's input.json
    with the true poses is missing, so the geometry is not consistent with the images: real polyline statistics,
    not a reconstruction."""

    def __init__(self, n_seeds):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import real_scene as rs
        self._sc, self._seeds, self.info = rs.real_edges_scene(n_seeds=n_seeds or 6268)
        self.scene = C.pointer(self._sc.c)
        self.seeds = C.pointer(self._seeds.c)
        self.n_seeds = int(self._seeds.c.n_seeds)
        self.n_views = int(self._sc.c.n_views)
        self.total_segments = int(self.info["segments_per_view"] * self.n_views)

    def seeds_np(self):
        return self._seeds.trk_off, self._seeds.trk_view, self._seeds.trk_xy
DESCR = {
    "c2": "C2 (BASELINE configs[1])",
    "c3": "C3' dtu006-shaped (BASELINE configs[2] with synthetic polylines and cameras)",
    "c4": "C4 (BASELINE configs[3])",
    "c3real": "C3-real: 25 real dtu006 edge maps -> polyline graphs (N2 builder), SYNTHETIC look-at cameras at the listed "
              "centres (the reference's input.json is missing: geometry not consistent with the images)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", choices=["auto", "c2", "c3", "c4", "c3real"], default="auto",
                    help="auto = c3 on one GPU, c4 (strong scaling) on several; c3real = the real dtu006 edge maps")
    ap.add_argument("--config", type=int, default=0, help="deprecated alias: 2/3/4 = --workload c2/c3/c4")
    ap.add_argument("--seeds", type=int, default=0, help="override the workload's seed count (experiments only)")
    ap.add_argument("--batch-seeds", type=int, default=0, help="seeds per step (default: all; c4: %d on one GPU, %d per rank on several)" % (C4_BATCH, C4_RANK_SHARE))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed side measurements (one step at a time, end to end): profiling passes want "
                         "exactly max(warmup, inflight) + steps passes of the hot path in the process")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the RCCL all-gather of the cloud even with one rank (exercises the N>1 code path)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="independent steps kept in flight per GPU, each on its own context/HIP stream driven by its "
                         "own host thread (1 = strictly one step at a time)")
    ap.add_argument("--path", choices=["refpoints", "sets"], default="refpoints",
                    help="refpoints = pipeline 3 (the headline path); sets = the pipelines 1-2 extractor (SURVEY N1) on "
                         "one synthetic polyline set per 3-D curve (single GPU only)")
    ap.add_argument("--cpu-runs", type=int, default=0, help="CPU baseline repetitions (default 5; c4: 1)")
    ap.add_argument("--cpu-seeds", type=int, default=0,
                    help="bound the CPU baseline to the first K seeds of the first step's batch (default: all; c4: 128)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON): libraries that print to file descriptor 1 (RCCL's
    # "Librccl path : ..." banner, for one) are sent to stderr; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # Several steps are kept in flight on separate HIP streams (plus the gather stream and RCCL's):
    # with the runtime's default of 4 hardware queues two of them can share a queue and serialise.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    # (Round 3: a context's chain working set is a fixed arena of ~0.5 GB — slots, not a slice per chain — plus the
    # staging area of one step's cloud, so nothing has to be capped per rank any more; the gathered cloud of a
    # 4096 x ranks step is held once, ~28 GB at 8 ranks.)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    from edgegraph3d_amd import api, host
    from edgegraph3d_amd.distributed import RcclCloudGather, shard_ranges_balanced

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, args.gpus))
    wl = args.workload
    if args.config in (2, 3, 4):
        wl = "c%d" % args.config
    if wl == "auto":
        wl = "c3" if world == 1 else "c4"
    dist = None
    # Dry run of the multi-rank control flow on a box with fewer GPUs than ranks (tests/bench_dryrun_check.sh): the
    # ranks share the visible GPUs, rendezvous over gloo, and the exchange step is replaced by a count reduction.
    # Not a measurement: the JSON line says so.
    dry = os.environ.get("EG3D_BENCH_DRYRUN_GATHER") == "1"
    if dry:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if api.device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if wl == "c3real":
        synth = _RealEdges(args.seeds)
    else:
        cfg = host.default_config(WORKLOADS[wl])
        if args.seeds:
            cfg.n_seeds = args.seeds
        synth = host.Synth(cfg)       # same seeded scene + seeds on every rank
    n_total = synth.n_seeds
    trk_off = synth.seeds_np()[0]
    batch = args.batch_seeds or ((C4_BATCH if world == 1 else C4_RANK_SHARE * world) if wl == "c4" else n_total)
    batch = min(batch, n_total)
    n_batches = max(1, n_total // batch)
    if args.path == "sets" and wl == "c3real":
        raise SystemExit("bench.py --path sets needs a synthetic workload (the sets come from its 3-D curves)")
    sets = synth.polyline_sets() if args.path == "sets" else None
    if sets is not None and world > 1:
        raise SystemExit("bench.py --path sets is a single-GPU measurement")

    def step_range(i):
        """Seed range of step i on this rank: batch i (cyclic) split into `world` contiguous,
        sum-of-track-length balanced ranges (rank order = seed order)."""
        b0 = (i % n_batches) * batch
        return shard_ranges_balanced(trk_off, b0, b0 + batch, world)[rank]

    inflight = max(1, args.inflight)

    # One context (own HIP stream, own work buffers) + one host thread per step in flight. ctypes
    # releases the GIL during the library call, so the threads really run concurrently.
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
            while True:
                job = self.todo.get()
                if job is None:
                    return
                step, device_only = job
                try:
                    if sets is not None:
                        self.done.put(self.ctx.match_polyline_sets(sets[0], sets[1], sets[2], device_only=device_only))
                    else:
                        b, e = step_range(step)
                        self.done.put(self.ctx.match_resident(b, e, device_only=device_only))
                except Exception as ex:  # surfaced by the main thread
                    self.done.put(ex)

    workers = [Worker()]
    workers += [Worker(workers[0]) for _ in range(inflight - 1)]

    gather = None
    if dist is not None:
        gstream = torch.cuda.Stream(device=dev, priority=-1)
        if dry:
            class _CountOnly:
                def allgather(self, local):
                    t = torch.tensor([int(local.n_points) if local is not None else 0, 0 if local is not None else 1],
                                     dtype=torch.int64)
                    dist.all_reduce(t)

                    class _C:
                        n_points = int(t[0].item())
                    return _C, (-4 if int(t[1].item()) else 0)

                def close(self):
                    pass
            gather = _CountOnly()
        else:
            gather = RcclCloudGather(dist, world, rank, local_rank, gstream.cuda_stream)

    def run_steps(first, n, pool, device_only=True):
        """Steps first..first+n-1, at most len(pool) in flight; results (and the collectives, which
        must be issued in the same order on every rank) are handled in step order by this thread."""
        out, submitted = [], 0
        for w in pool[:n]:
            w.todo.put((first + submitted, device_only))
            submitted += 1
        for i in range(n):
            w = pool[i % len(pool)]
            r = w.done.get()
            failed = r if isinstance(r, Exception) else None
            if gather is not None:
                # a rank whose step failed still takes part in the collective (local = None): the status word makes
                # EVERY rank return EG3D_GATHER_ERR_INCOMPLETE, so no rank is left blocked in the exchange
                cloud, rc = gather.allgather(None if failed else w.ctx.last_device_output())  # synchronous on the gather stream
                if failed is not None:
                    raise failed
                if rc != 0:
                    raise RuntimeError("eg3d_allgather_edgepoints failed on every rank with rc=%d" % rc)
                total = int(cloud.n_points)
            else:
                if failed is not None:
                    raise failed
                total = r["n_points"]
            if submitted < n:
                w.todo.put((first + submitted, device_only))
                submitted += 1
            out.append((r, total))
        return out

    run_steps(0, max(args.warmup, inflight), workers)  # every context sizes its buffers before the timed region
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    results = run_steps(0, args.steps, workers)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dry else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_ms = {k: [r["times"][k] for r, _ in results] for _, k in STAGES}
    points_done = sum(tot for _, tot in results)
    last, _ = results[-1]
    bytes_alg = sum(r["times"]["bytes_algorithmic"] for r, _ in results) / len(results)  # this rank, per step
    # the same steps strictly one at a time, and end to end (with the D2H copy of the cloud): untimed
    # side measurements reported beside the value
    single = e2e = None
    if rank == 0 and world == 1 and not args.no_extras:
        ns = max(3, min(10, args.steps))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        rs = run_steps(0, ns, workers[:1])
        torch.cuda.synchronize()
        single = ((time.perf_counter() - ts) / ns, sum(tot for _, tot in rs) / ns,
                  {k: sum(r["times"][k] for r, _ in rs) / ns for _, k in STAGES})
        # timed at the C ABI (the Python wrapper's numpy conversion is not part of the product)
        if sets is None:
            tt = [workers[0].ctx.time_match_to_host(*step_range(i)) for i in range(ns)]
        else:
            tt = [workers[0].ctx.time_match_sets_to_host(sets[0], sets[1], sets[2]) for _ in range(ns)]
        e2e = (sum(t for t, _ in tt) / ns, sum(n for _, n in tt) / ns)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = points_done / elapsed
        avg = {k: (sum(v) / len(v) if v else 0.0) for k, v in stage_ms.items()}
        # the dominant kernel is k3b_expand on every workload profiled (profiles/r02_*); its stage is that one
        # kernel, so the HIP-event time below is comparable with rocprofv3's per-kernel average
        dom_name, dom_key = next(s for s in STAGES if s[0] == "k3b_expand")
        dom_ms = avg[dom_key]
        achieved = (bytes_alg / (dom_ms * 1e-3)) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        wkey = wl if sets is None else wl + "_sets"
        if os.path.exists(pmc):
            try:
                ent = json.load(open(pmc)).get(wkey, {})
                traffic = ent.get(dom_name, {}).get("hbm_bytes_per_step")
                traffic_src = ent.get("provenance")
            except Exception:
                traffic = None
        V = synth.n_views
        segs = synth.total_segments / V
        if sets is None:
            workload = ("%s: %d views / %d seeds / %.0f polyline segments per view; one step = %d seeds%s"
                        % (DESCR[wl], V, n_total, segs, batch,
                           "" if n_batches == 1 else " (steps cycle through %d batches)" % n_batches))
        else:
            workload = ("%s, pipelines 1-2 extractor: %d views / %d polyline sets (%d polylines) / %.0f segments per view"
                        % (DESCR[wl], V, sets[0], len(sets[2]), segs))
        line = {
            "metric": "triangulated edge-points/sec", "value": value, "unit": "edge-points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "DRY RUN of the multi-rank control flow (ranks share a GPU, no collective): not a measurement" if dry else
                    "synthetic" if wl != "c3real" else "real dtu006 edge maps + synthetic cameras and seeds", "steps_in_flight": inflight,
            "config": {
                "workload": workload, "workload_key": wkey,
                "edge_points_per_step": points_done / args.steps, "observations_last_step_rank0": int(last["n_obs"]),
                "tasks_last_step_rank0": int(last["n_tasks"]), "hypotheses_last_step_rank0": int(last["n_hypotheses"]),
                "chains_last_step_rank0": int(last["n_chains"]),
                "parallelism": ("1 GPU" if world == 1 else
                                "%d ranks: whole scene per rank, each step's seeds split into contiguous ranges balanced "
                                "by the sum of track lengths, RCCL all-gather of the cloud (eg3d_allgather_edgepoints)"
                                % world),
                "arithmetic": "2-D geometry f32, DLT+Gauss-Newton f64 (as the reference)",
                "dlt_form": ("6x4 system of cv::triangulatePoints in OpenCV <= 3.1 (the release the reference names)"
                             if api.lib().eg3d_dlt_rows() == 3 else "4x4 system of cv::triangulatePoints in OpenCV >= 3.2 "
                             "(EG3D_LIB = libeg3d_dlt4x4.so)"),
            },
            "stage_ms": {n: round(avg[k], 4) for n, k in STAGES},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_step": int(bytes_alg), "kernel_ms_per_step": dom_ms,
                         "note": "per step of rank 0: algorithmic bytes of the step / the kernel's HIP-event time in that "
                                 "step (one launch per step unless the scratch budget forces chunks); with several "
                                 "steps in flight the launches of different steps share the GPU, so this duration is "
                                 "longer than the kernel's exclusive time (see roofline.exclusive)"},
        }
        if world > 1:
            # the driver derives scaling efficiency itself; this is only where the 1-GPU figure of the SAME workload
            # lives (the default 1-GPU run measures C3')
            ref = os.path.join(ROOT, "profiles", "r03_final_%s.json" % wkey)
            if os.path.exists(ref):
                try:
                    r1 = json.load(open(ref))
                    line["same_workload_on_1_gpu"] = {"value": r1["value"], "ms_per_step": r1["ms_per_step"],
                                                      "source": "profiles/r03_final_%s.json (python bench.py --workload %s; one "
                                                                "GPU takes the same seeds in steps of %d)" % (wkey, wkey, C4_BATCH)}
                except Exception:
                    pass
        if single is not None:
            line["ms_per_step_one_at_a_time"] = single[0] * 1e3
            line["stage_ms_one_at_a_time"] = {n: round(single[2][k], 4) for n, k in STAGES}
            ex_ms = single[2][dom_key]
            if ex_ms > 0:
                ex = bytes_alg / (ex_ms * 1e-3) / 1e9
                line["roofline"]["exclusive"] = {"kernel_ms_per_step": ex_ms, "achieved": ex, "frac": ex / HBM_PEAK_GBS,
                                                 "what": "the same kernel with one step on the GPU at a time"}
            line["value_one_step_at_a_time"] = single[1] / single[0]
            line["end_to_end"] = {"ms_per_step": e2e[0] * 1e3, "value": e2e[1] / e2e[0],
                                  "what": "one step at a time incl. the D2H copy of the edge-point cloud into "
                                          "caller-owned host arrays (eg3d_match_resident / eg3d_match_polyline_sets, device_only=0), timed at the C ABI"}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import binding as ob   # cpu_baseline leg: the checker timed as the CPU port
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from parity_util import compare_edgepoints
            orc = ob.Oracle(synth.scene)
            if sets is not None:
                def cpu_run(lo, hi, nthreads):
                    return orc.match_polyline_sets(sets[0], sets[1], sets[2], lo, hi, nthreads)
                b, e = 0, sets[0]
                unit = "sets"
            else:
                def cpu_run(lo, hi, nthreads):
                    return orc.match(synth.seeds, lo, hi, nthreads)
                b, e = step_range(0)
                unit = "seeds"
            k = args.cpu_seeds or (128 if wl == "c4" else 0)
            ce = e if not k else min(e, b + k)
            runs = args.cpu_runs or (1 if wl == "c4" else 5)
            cpu_run(b, min(ce, b + max(1, (ce - b) // 20)), 1)  # warm-up
            secs, pts = [], 0
            for _ in range(runs):
                r = cpu_run(b, ce, 1)
                secs.append(r["stats"]["seconds"])
                pts = r["n_points"]
            med = statistics.median(secs)
            ncores = os.cpu_count() or 1
            rall = cpu_run(b, ce, ncores)
            line["cpu_baseline"] = {
                "value": pts / med, "unit": "edge-points/s", "cores": 1, "kind": "port",
                "sample": "%s of step 0 (%d %s, %d edge-points), oracle g++ -O3, 1 thread, median of %d run(s) "
                          "(%.2f s each: %s); scene/grid construction excluded"
                          % ("all" if ce == e else "first %d %s" % (ce - b, unit), ce - b, unit, pts, len(secs), med,
                             ", ".join("%.2f" % s for s in secs)),
                "all_cores": {"value": rall["n_points"] / rall["stats"]["seconds"], "cores": ncores},
                "oracle_algorithmic_bytes": int(r["stats"]["bytes_algorithmic"]),
            }
            line["speedup_vs_cpu_1thread"] = value / (pts / med)
            if single is not None:
                line["speedup_vs_cpu_1thread_one_step_at_a_time"] = (single[1] / single[0]) / (pts / med)
            # "CPU-ref parity err" (the second half of BASELINE.json's metric): the GPU output of the same
            # seeds / sets copied to the host and compared with the oracle's — relative error of the 3-D
            # coordinates (north star: <= 1e-4) and exactness of every id / view list and of the order
            if sets is not None:
                gfull = workers[0].ctx.match_polyline_sets(sets[0], sets[1], sets[2], b, ce)
            else:
                gfull = workers[0].ctx.match_resident(b, ce)
            rep = compare_edgepoints(r, gfull, rel_tol=1e-4)
            line["parity"] = {"vs": "oracle (CPU restatement, same DLT form; parity unpinned, see DESIGN.md 3)", "points_compared": int(pts),
                              "max_rel_err_X": rep.get("max_rel_X"), "X_bit_exact": rep.get("bitexact_X"),
                              "ids_views_order_exact": bool(rep["ok"]), "obs_xy_bit_exact": rep.get("bitexact_xy"),
                              "tolerance": 1e-4}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    for w in workers:
        w.todo.put(None)
        w.join(timeout=10)
        w.ctx.close()
    if gather is not None:
        gather.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        # worker threads may still be inside the library: report and leave without running destructors under them
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
