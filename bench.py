#!/usr/bin/env python3
"""bench.py — throughput of the refpoint -> epipolar match -> triangulate hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload auto|c2|c3|c4|c3real]

A "step" = one pass of the hot path (eg3d_match_resident: K1..K4) over one batch of synthetic
seeds whose scene and tracks are already resident in HBM.

  N = 1 (default workload c3): C3' = the dtu006-shaped configuration the north-star target is
      quoted on (25 views / 6268 seeds / ~15k polyline segments per view; BASELINE configs[2] with
      synthetic polylines — the reference's input.json for the real images is missing). One step
      = all 6268 seeds. The same line carries two sub-measurements made after the timed region:
      `scaling_base` = a short leg of the multi-GPU workload (C4, steps of 8192 seeds) on this one
      GPU, so that the N = 1 point of the scaling curve is measured by the same command, and `c5` =
      BASELINE configs[4] (Gauss-Newton filter over 1 M points).
  N > 1 (default workload c4, launched by torch.distributed.run, one rank per GPU): BASELINE
      configs[3], 200 views / 100 000 seeds / ~20k segments per view, STRONG scaling: one step =
      one batch of 4096 x N seeds (steps cycle through the whole batches of the 100k seeds), split
      over the ranks into contiguous ranges balanced by the sum of track lengths; every rank holds
      the whole scene; the step ends with the RCCL all-gather of the edge-point cloud through the
      C ABI (eg3d_allgather_edgepoints, include/eg3d_rccl.h). `time_to_solution_s` = ONE pass over
      all 100 000 seeds (last, partial batch included) with the gathers, max over ranks.

Rank 0 prints ONE JSON line. value = whole-job edge-points per second with `steps_in_flight`
independent steps overlapped per GPU (K steps are still exactly K passes); the same steps strictly
one at a time are reported beside it (value_one_step_at_a_time / ms_per_step_one_at_a_time), and
`end_to_end` adds the D2H copy of the cloud. `roofline` = dominant kernel against the HBM peak:
algorithmic bytes of SURVEY 8(d) / that kernel's HIP-event time with ONE step on the GPU at a time
(the figure rocprofv3's per-kernel average reproduces; the overlapped in-flight duration is under
roofline.in_flight); traffic = PMC bytes from the committed rocprofv3 passes of this workload, with
provenance. `roofline_valu` = the same kernel against the vector-ALU issue peak (what actually
bounds the path). `cpu_baseline` = the CPU oracle, 1 thread, on this box (N=1 only).
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
CLOCK_HZ = 2.4e9       # same guide: max shader clock
N_SIMD = 1024          # 256 CUs x 4 SIMDs, 16 lanes issued per SIMD per cycle
C4_BATCH = 8192        # seeds per step of the c4 workload on one GPU (12 whole batches in 100 000 seeds)
C4_RANK_SHARE = 4096   # ... and per RANK on several: a step's batch = 4096 x ranks seeds. A chain of this workload runs
                       # for up to ~0.45 s on its own wavefront, so a rank needs a few thousand seeds per step to keep its
                       # GPU full behind such tails (measured on one GPU, 4 steps in flight: 1024 / 2048 / 4096 / 8192 seeds
                       # per step -> 0.98 / 1.70 / 1.94 / 1.98 M edge-points/s). The job (100 000 seeds) is the same at every N.
STAGES = [("k1_seed_candidates", "ms_candidates"), ("k2_epipolar_hits", "ms_epipolar"),
          ("k3a_hypotheses", "ms_hypotheses"), ("k3s_select", "ms_select"), ("k3b_expand", "ms_expand"),
          ("k4_emit", "ms_emit")]
DOM_NAME, DOM_KEY = "k3b_expand", "ms_expand"  # the dominant kernel on every workload profiled (profiles/)
WORKLOADS = {"c2": 2, "c3": 3, "c4": 4}
CLOUD_BYTES_PER_POINT = 12 + 8 + 16    # X, obs_off, key
CLOUD_BYTES_PER_OBS = 4 + 4 + 4 + 8    # view, polyline, segment, xy


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


def _overlap_fraction(gathers, computes):
    """Fraction of the time spent inside the gather calls during which at least one step of this rank was being
    computed (host clock: a step's interval is its library call in the worker thread)."""
    total = sum(e - b for b, e in gathers)
    if total <= 0:
        return None
    ev = sorted(computes)
    hidden = 0.0
    for gb, ge in gathers:
        cur = gb
        for cb, ce in ev:
            if ce <= cur or cb >= ge:
                continue
            lo, hi = max(cb, cur), min(ce, ge)
            if hi > lo:
                hidden += hi - lo
                cur = hi
    return hidden / total


class Leg:
    """One workload measured the way the contract asks: contexts + host threads for the steps in flight, warm-up,
    a timed region of exactly K steps between barriers and synchronisations, and (optionally) the side measurements."""

    def __init__(self, env, wl, path="refpoints", seeds=0, batch_seeds=0, inflight=4):
        import numpy as np  # noqa: F401
        from edgegraph3d_amd import api, host
        from edgegraph3d_amd.distributed import StepPlan
        self.env, self.wl, self.api = env, wl, api
        world = env["world"]
        if wl == "c3real":
            self.synth = _RealEdges(seeds)
        else:
            cfg = host.default_config(WORKLOADS[wl])
            if seeds:
                cfg.n_seeds = seeds
            self.synth = host.Synth(cfg)       # same seeded scene + seeds on every rank
        self.n_total = self.synth.n_seeds
        self.trk_off = self.synth.seeds_np()[0]
        batch = batch_seeds or ((C4_BATCH if world == 1 else C4_RANK_SHARE * world) if wl == "c4" else self.n_total)
        self.plan = StepPlan(self.trk_off, self.n_total, batch, world, env["rank"])   # the per-step sharding (also what tests/test_multirank_gloo.py runs)
        self.batch, self.n_batches = self.plan.batch, self.plan.n_batches
        if path == "sets" and wl == "c3real":
            raise SystemExit("bench.py --path sets needs a synthetic workload (the sets come from its 3-D curves)")
        self.sets = self.synth.polyline_sets() if path == "sets" else None
        if self.sets is not None and world > 1:
            raise SystemExit("bench.py --path sets is a single-GPU measurement")
        self.inflight = max(1, inflight)
        leg = self

        # One context (own HIP stream, own work buffers) + one host thread per step in flight. ctypes
        # releases the GIL during the library call, so the threads really run concurrently.
        class Worker(threading.Thread):
            def __init__(self, parent=None):
                super().__init__(daemon=True)
                if parent is None:
                    t0 = time.perf_counter()
                    self.ctx = api.Context(leg.synth.scene, env["local_rank"])
                    t1 = time.perf_counter()
                    self.ctx.upload_seeds(leg.synth.seeds)   # inputs resident in HBM before the timed region
                    leg.setup_ms = {"eg3d_create": (t1 - t0) * 1e3, "eg3d_upload_seeds": (time.perf_counter() - t1) * 1e3,
                                    "what": "this leg's first context (the first eg3d_create of a process also pays the HIP "
                                            "runtime's start-up, ~0.1 s: see time_to_solution_one_shot for a warm one)"}
                else:
                    self.ctx = parent.ctx.clone()        # shares the resident scene and seeds (eg3d_clone)
                self.todo, self.done = queue.Queue(), queue.Queue()
                self.start()

            def run(self):
                while True:
                    job = self.todo.get()
                    if job is None:
                        return
                    (b, e), device_only = job
                    t0 = time.perf_counter()
                    try:
                        if leg.sets is not None:
                            r = self.ctx.match_polyline_sets(leg.sets[0], leg.sets[1], leg.sets[2], device_only=device_only)
                        else:
                            r = self.ctx.match_resident(b, e, device_only=device_only)
                        r["_interval"] = (t0, time.perf_counter())
                        self.done.put(r)
                    except Exception as ex:  # surfaced by the main thread
                        self.done.put(ex)

        self.workers = [Worker()]
        self.workers += [Worker(self.workers[0]) for _ in range(self.inflight - 1)]
        self.gather_log = []   # (begin, end, bytes received by this rank) per collective
        self.compute_log = []  # (begin, end) per step of this rank

    def step_range(self, i):
        """Seed range of step i on this rank: batch i (cyclic over the WHOLE batches) split into `world` contiguous,
        sum-of-track-length balanced ranges (rank order = seed order)."""
        return self.plan.step_range(i)

    def pass_range(self, i):
        """The same for ONE pass over all seeds: ceil(n / batch) steps, the last one partial."""
        return self.plan.pass_range(i)

    def n_pass_steps(self):
        return self.plan.n_pass_steps()

    def run_steps(self, first, n, pool, device_only=True, ranges=None):
        """Steps first..first+n-1, at most len(pool) in flight; results (and the collectives, which
        must be issued in the same order on every rank) are handled in step order by this thread."""
        gather = self.env["gather"]
        ranges = ranges or self.step_range
        out, submitted = [], 0
        for w in pool[:n]:
            w.todo.put((ranges(first + submitted), device_only))
            submitted += 1
        for i in range(n):
            w = pool[i % len(pool)]
            r = w.done.get()
            failed = r if isinstance(r, Exception) else None
            if failed is None:
                self.compute_log.append(r.pop("_interval"))
            if gather is not None:
                # a rank whose step failed still takes part in the collective (local = None): the status word makes
                # EVERY rank return EG3D_GATHER_ERR_INCOMPLETE, so no rank is left blocked in the exchange
                g0 = time.perf_counter()
                cloud, rc = gather.allgather(None if failed else w.ctx.last_device_output())  # synchronous on the gather stream
                g1 = time.perf_counter()
                if failed is not None:
                    raise failed
                if rc != 0:
                    raise RuntimeError("eg3d_allgather_edgepoints failed on every rank with rc=%d" % rc)
                total = int(cloud.n_points)
                recv_pts = total - int(r["n_points"])
                recv_obs = int(getattr(cloud, "n_obs", 0)) - int(r["n_obs"])
                self.gather_log.append((g0, g1, max(0, recv_pts) * CLOUD_BYTES_PER_POINT + max(0, recv_obs) * CLOUD_BYTES_PER_OBS))
            else:
                if failed is not None:
                    raise failed
                total = r["n_points"]
            if submitted < n:
                w.todo.put((ranges(first + submitted), device_only))
                submitted += 1
            out.append((r, total))
        return out

    def timed(self, steps, warmup):
        """The contract's timed region: W warm-up steps (at least one per context), then exactly K steps between
        barrier + synchronize on both sides; the MAX over ranks is returned."""
        import torch
        dist = self.env["dist"]
        self.run_steps(0, max(warmup, self.inflight), self.workers)  # every context sizes its buffers before the timed region
        self.gather_log.clear()
        self.compute_log.clear()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        results = self.run_steps(0, steps, self.workers)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return self._max_over_ranks(time.perf_counter() - t0), results

    def _max_over_ranks(self, seconds):
        import torch
        dist = self.env["dist"]
        if dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if self.env["dry"] else self.env["dev"])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather_stats(self):
        if not self.gather_log:
            return None
        ms = [(e - b) * 1e3 for b, e, _ in self.gather_log]
        return {"gather_ms": sum(ms) / len(ms), "gather_ms_max": max(ms),
                "gather_bytes": sum(x for _, _, x in self.gather_log) / len(self.gather_log),
                "overlap_frac": _overlap_fraction([(b, e) for b, e, _ in self.gather_log], self.compute_log),
                "what": "per step on rank 0: wall time inside eg3d_allgather_edgepoints (counts + grouped send/recv + offset "
                        "rebase, synchronous on the gather stream), bytes this rank RECEIVED, and the fraction of that time "
                        "during which another step of this rank was being computed (steps in flight)"}

    def time_to_solution(self):
        """ONE pass over all seeds of the workload (every batch once, the last one partial), gathers included, the
        contexts warm: barrier + synchronize on both sides, max over ranks."""
        import torch
        dist = self.env["dist"]
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rs = self.run_steps(0, self.n_pass_steps(), self.workers, ranges=self.pass_range)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        return self._max_over_ranks(time.perf_counter() - t0), sum(tot for _, tot in rs)

    def serial_and_e2e(self, steps):
        """The same steps strictly one at a time, and end to end (with the D2H copy of the cloud)."""
        import torch
        ns = max(3, min(10, steps))
        torch.cuda.synchronize()
        ts = time.perf_counter()
        rs = self.run_steps(0, ns, self.workers[:1])
        torch.cuda.synchronize()
        single = ((time.perf_counter() - ts) / ns, sum(tot for _, tot in rs) / ns,
                  {k: sum(r["times"][k] for r, _ in rs) / ns for _, k in STAGES})
        # timed at the C ABI (the Python wrapper's numpy conversion is not part of the product)
        ctx = self.workers[0].ctx
        # (one untimed call first: a host call runs on the context's internal lanes, which are created — with their work
        # buffers and pinned staging — by the first call that needs them; the one-shot cost is under time_to_solution_one_shot)
        if self.sets is None:
            ctx.time_match_to_host(*self.step_range(0))
            ctx.time_match_to_host(*self.step_range(0))   # (the second host call of a context is the one that creates its lanes)
            tt = [ctx.time_match_to_host(*self.step_range(i)) for i in range(ns)]
        else:
            ctx.time_match_sets_to_host(self.sets[0], self.sets[1], self.sets[2])
            ctx.time_match_sets_to_host(self.sets[0], self.sets[1], self.sets[2])
            tt = [ctx.time_match_sets_to_host(self.sets[0], self.sets[1], self.sets[2]) for _ in range(ns)]
        ts = sorted(t for t, _ in tt)
        self.e2e_spread = {"min_ms": ts[0] * 1e3, "max_ms": ts[-1] * 1e3, "mean_ms": sum(ts) / ns * 1e3, "calls": ns}
        return single, (statistics.median(ts), sum(n for _, n in tt) / ns)

    def setup_and_time_to_solution(self):
        """What a one-shot caller pays (setup reported separately): a FRESH context — eg3d_create = scene
        validation, bounding boxes, the 2 x V uniform grids (built on host threads), uploads —, eg3d_upload_seeds, and ONE
        eg3d_match_resident call with the D2H copy of the cloud, nothing warm (work buffers, lanes and pinned staging are
        allocated inside the call); then the same call again on the now warm context. Seconds, timed at the C ABI."""
        api = self.api
        b, e = self.step_range(0)
        t0 = time.perf_counter()
        ctx = api.Context(self.synth.scene, self.env["local_rank"])
        t1 = time.perf_counter()
        ctx.upload_seeds(self.synth.seeds)
        t2 = time.perf_counter()
        cold, n = ctx.time_match_to_host(b, e)
        warm = min(ctx.time_match_to_host(b, e)[0] for _ in range(3))
        # a second create on the same process (driver state warm): the figure the grid construction itself costs
        t3 = time.perf_counter()
        ctx2 = api.Context(self.synth.scene, self.env["local_rank"])
        t4 = time.perf_counter()
        ctx2.close()
        ctx.close()
        return {"create_ms": (t1 - t0) * 1e3, "create_again_ms": (t4 - t3) * 1e3, "upload_seeds_ms": (t2 - t1) * 1e3,
                "first_call_ms": cold * 1e3, "warm_call_ms": warm * 1e3, "edge_points": n,
                "time_to_solution_ms": (t2 - t0 + cold) * 1e3,
                "what": "fresh context: eg3d_create (validation, bounding boxes, 2 x V grids on host threads, uploads) + "
                        "eg3d_upload_seeds + ONE eg3d_match_resident call with the D2H copy into caller-owned arrays, work "
                        "buffers cold (first_call_ms; warm_call_ms = best of 3 more calls on the same context); "
                        "create_again_ms = a second eg3d_create of the same scene in this process"}

    def describe(self):
        V = self.synth.n_views
        segs = self.synth.total_segments / V
        if self.sets is None:
            return ("%s: %d views / %d seeds / %.0f polyline segments per view; one step = %d seeds%s"
                    % (DESCR[self.wl], V, self.n_total, segs, self.batch,
                       "" if self.n_batches == 1 else " (steps cycle through %d batches)" % self.n_batches))
        return ("%s, pipelines 1-2 extractor: %d views / %d polyline sets (%d polylines) / %.0f segments per view"
                % (DESCR[self.wl], V, self.sets[0], len(self.sets[2]), segs))

    def close(self):
        for w in self.workers:
            w.todo.put(None)
            w.join(timeout=10)
            w.ctx.close()
        self.workers = []


def _usable_cores():
    """(cores this process can really use, how that was found). os.cpu_count() is what the box SHOWS; a container's CPU
    quota (cgroup v2 cpu.max / v1 cfs quota) and the affinity mask are what it GETS — on the GPU boxes of this project 256
    threads are visible and the quota is 16 cores: the oracle scales x15.5 up to 16 threads and gets slower beyond
    (profiles/r05_cpu_scaling.json), so "all host cores" of the CPU comparator means the quota."""
    import math
    n = os.cpu_count() or 1
    how = ["os.cpu_count() = %d" % n]
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n = a
        how.append("affinity %d" % a)
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            c = max(1, int(math.ceil(float(q) / float(p))))
            how.append("cgroup cpu.max %s/%s = %d cores" % (q, p, c))
            n = min(n, c)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                c = max(1, int(math.ceil(q / p)))
                how.append("cgroup cfs quota %d/%d = %d cores" % (q, p, c))
                n = min(n, c)
        except Exception:
            pass
    return n, "; ".join(how)


def _pmc_entry(wkey):
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(pmc):
        return {}
    try:
        return json.load(open(pmc)).get(wkey, {})
    except Exception:
        return {}


def _rooflines(wkey, bytes_alg, excl_ms, inflight_ms):
    """roofline (HBM, the contract's object) and roofline_valu for the dominant kernel. Primary figures = the kernel with
    ONE step on the GPU at a time when that was measured (so that kernel time <= the serial step time); with several
    steps in flight the launches of different steps share the GPU and the event duration is an overlapped wall time."""
    ent = _pmc_entry(wkey)
    dom = ent.get(DOM_NAME, {})
    # the committed PMC figures belong to the device sources they were measured on: a changed kernel makes them stale
    stale = None
    try:
        from edgegraph3d_amd import build as _build
        fp_now, fp_prof = _build.device_source_fingerprint(), ent.get("source_fingerprint")
        env_set = _build.kernel_choice_env_set()
        if env_set:
            stale = {"profiled_sources": fp_prof, "these_sources": fp_now, "last_measured_traffic": dom.get("hbm_bytes_per_step"),
                     "what": "environment switches that choose the kernel build or its launch are set (%s): the committed PMC pass "
                             "does not describe this run" % ", ".join(env_set)}
        elif dom and not fp_prof:
            stale = {"profiled_sources": None, "these_sources": fp_now, "last_measured_traffic": dom.get("hbm_bytes_per_step"),
                     "what": "the committed PMC pass of this workload predates the source fingerprint (rounds 2-4): not reported as this run's traffic"}
        elif fp_prof and fp_prof != fp_now:
            stale = {"profiled_sources": fp_prof, "these_sources": fp_now, "last_measured_traffic": dom.get("hbm_bytes_per_step"),
                     "what": "the committed PMC pass was made on other device sources / switches than this build: not reported as this run's traffic; re-run tools/profile_round.sh"}
    except Exception:
        pass
    primary_ms, basis = (excl_ms, "one step on the GPU at a time") if excl_ms else (inflight_ms, "steps in flight (no exclusive measurement in this run)")
    ach = (bytes_alg / (primary_ms * 1e-3)) / 1e9 if primary_ms and primary_ms > 0 else 0.0
    roof = {"bound": "hbm", "kernel": DOM_NAME, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None if stale else dom.get("hbm_bytes_per_step"), "traffic_source": ent.get("provenance"),
            "algorithmic_bytes_per_step": int(bytes_alg), "kernel_ms_per_step": primary_ms, "measured_with": basis,
            "note": "per step of rank 0: algorithmic bytes of the step (SURVEY 8d) / the kernel's HIP-event time in that step "
                    "(one launch per step)"}
    if stale:
        roof["traffic_stale"] = stale
    if excl_ms and inflight_ms:
        a2 = (bytes_alg / (inflight_ms * 1e-3)) / 1e9
        roof["in_flight"] = {"kernel_ms_per_step": inflight_ms, "achieved": a2, "frac": a2 / HBM_PEAK_GBS,
                             "what": "the same kernel's event duration inside the timed region, where the launches of the "
                                     "steps in flight overlap: a wall time, not a per-step cost"}
    valu = None
    v = dom.get("valu")
    if v and primary_ms:
        # thread-cycles of VALU work per launch (quad-cycle units) against 1024 SIMDs x 64 lane-slots per quad-cycle
        prof_ms = v.get("kernel_ms") or primary_ms
        peak_prof = N_SIMD * 64.0 * (prof_ms * 1e-3 * CLOCK_HZ / 4.0)
        valu = {"bound": "valu", "kernel": DOM_NAME, "unit": "lane-slots (quad-cycle units) per launch",
                "achieved": v.get("SQ_THREAD_CYCLES_VALU"), "peak": peak_prof,
                "frac": (v.get("SQ_THREAD_CYCLES_VALU") or 0.0) / peak_prof if peak_prof else None,
                "valu_busy_frac": (v.get("SQ_ACTIVE_INST_VALU") or 0.0) / (N_SIMD * (prof_ms * 1e-3 * CLOCK_HZ / 4.0)),
                "active_lane_frac": v.get("active_lane_frac"), "kernel_ms_in_profile": prof_ms,
                "source": ent.get("provenance_valu") or ent.get("provenance"),
                "what": "SQ_THREAD_CYCLES_VALU (active lanes x issue time of every vector-ALU instruction) / (1024 SIMDs x 64 "
                        "lane-slots x quad-cycles of the kernel's duration at 2.4 GHz), from the committed PMC pass of this "
                        "workload: the share of the chip's vector-ALU lane-slots that did work — frac = valu_busy_frac x "
                        "active_lane_frac. The path is bound here, not by HBM."}
    return roof, valu


def _c5_subline(device):
    """BASELINE configs[4]: gaussNewtonFiltering over 1 M synthetic points (k5_gn_filter), kernel HIP-event time,
    bit-exact parity of ALL points against the oracle, the oracle timed beside it on a sample."""
    import numpy as np
    from edgegraph3d_amd import api, host
    n = 1000000
    s = host.Synth(5)   # 16-view rig: k really spans 3..10
    X, off, view, xy = s.points(n)
    ctx = api.Context(s.scene, device)
    ms, wall = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        Xo, inl, m = ctx.gn_filter(X, off, view, xy, 2.25)
        wall.append(time.perf_counter() - t0)
        ms.append(m)
    k_ms = float(statistics.median(ms[1:]))
    e2e_ms = 1e3 * float(statistics.median(wall[1:]))
    n_obs = int(off[-1])
    alg = n * (12 + 12 + 1 + 4) + n_obs * 12  # X in/out, inlier, obs_off; per observation view id + xy
    out = {"workload": "C5 (BASELINE configs[4]): %d points, %d observations (k~U[3,10], mean %.2f), 16-view rig, gn_max_mse 2.25"
                       % (n, n_obs, n_obs / n),
           "kernel": "k5_gn_filter", "kernel_ms": k_ms, "value": n / (k_ms * 1e-3), "unit": "points/s", "dtype": "f32",
           "inlier_frac": float(inl.mean()),
           "end_to_end": {"ms": e2e_ms, "value": n / (e2e_ms * 1e-3), "unit": "points/s",
                          "what": "one eg3d_gn_filter call from caller-owned host arrays to caller-owned host arrays: input checks, H2D of "
                                  "the points and observations (%.0f MB, pageable), k5_gn_filter, D2H of X and the inlier flags (%.0f MB); "
                                  "wrapper's numpy conversions included" % ((n * 16 + n_obs * 12) / 1e6, n * 13 / 1e6)},
           "roofline": {"bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg,
                        "note": "FP32/FP64-VALU-bound, not HBM-bound: see roofline_valu (profiles/r06_c5_rocprof_summary.txt; "
                                "iteration statistics profiles/r06_c5_iterations.json)"}}
    try:
        v = json.load(open(os.path.join(ROOT, "profiles", "r06_c5_valu.json")))
        prof_ms = v.get("kernel_ms") or k_ms
        peak = N_SIMD * 64.0 * (prof_ms * 1e-3 * CLOCK_HZ / 4.0)
        out["roofline_valu"] = {"bound": "valu", "kernel": "k5_gn_filter", "unit": "lane-slots (quad-cycle units) per launch",
                                "achieved": v.get("SQ_THREAD_CYCLES_VALU"), "peak": peak,
                                "frac": (v.get("SQ_THREAD_CYCLES_VALU") or 0.0) / peak,
                                "valu_busy_frac": (v.get("SQ_ACTIVE_INST_VALU") or 0.0) / (N_SIMD * (prof_ms * 1e-3 * CLOCK_HZ / 4.0)),
                                "active_lane_frac": v.get("active_lane_frac"), "kernel_ms_in_profile": prof_ms,
                                "source": "profiles/r06_c5_rocprof_summary.txt (committed PMC pass of tools/profile_c5.sh)"}
    except Exception:
        pass
    from oracle import binding as ob   # checker + cpu_baseline leg
    o = ob.Oracle(s.scene)
    m = 200000
    t = time.time()
    Xr, ir = o.gn_filter(X[:m], off[:m + 1], view[:off[m]], xy[:off[m]], 2.25, nthreads=1)
    dt = time.time() - t
    out["cpu_baseline"] = {"value": m / dt, "unit": "points/s", "cores": 1, "kind": "port",
                           "sample": "first %d points, oracle g++ -O3, 1 thread, %.2f s" % (m, dt)}
    Xa, ia = o.gn_filter(X, off, view, xy, 2.25, nthreads=os.cpu_count() or 1)
    out["parity"] = {"points_compared": n, "X_bit_exact": bool(np.array_equal(Xo.view(np.uint32), Xa.view(np.uint32))),
                     "inlier_flags_exact": bool(np.array_equal(inl, ia))}
    ctx.close()
    return out


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
                    help="skip the untimed side measurements (one step at a time, end to end, time to solution, the C4 and C5 "
                         "sub-lines): profiling passes want exactly max(warmup, inflight) + steps passes of the hot path in the process")
    ap.add_argument("--no-sublines", action="store_true", help="skip only the scaling_base (C4) and c5 sub-measurements")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the RCCL all-gather of the cloud even with one rank (exercises the N>1 code path)")
    ap.add_argument("--inflight", type=int, default=4,
                    help="independent steps kept in flight per GPU, each on its own context/HIP stream driven by its "
                         "own host thread (1 = strictly one step at a time)")
    ap.add_argument("--path", choices=["refpoints", "sets"], default="refpoints",
                    help="refpoints = pipeline 3 (the headline path); sets = the pipelines 1-2 extractor (SURVEY N1) on "
                         "one synthetic polyline set per 3-D curve (single GPU only)")
    ap.add_argument("--cpu-runs", type=int, default=0, help="timed CPU baseline runs after the warm-up (default 5, median; ~5 s each on the C3' sample)")
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
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    from edgegraph3d_amd import api
    from edgegraph3d_amd.distributed import RcclCloudGather

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

    gather = None
    if dist is not None:
        gstream = torch.cuda.Stream(device=dev, priority=-1)
        if dry:
            class _CountOnly:
                def allgather(self, local):
                    t = torch.tensor([int(local.n_points) if local is not None else 0, 0 if local is not None else 1,
                                      int(local.n_obs) if local is not None else 0], dtype=torch.int64)
                    dist.all_reduce(t)

                    class _C:
                        n_points = int(t[0].item())
                        n_obs = int(t[2].item())
                    return _C, (-4 if int(t[1].item()) else 0)

                def close(self):
                    pass
            gather = _CountOnly()
        else:
            gather = RcclCloudGather(dist, world, rank, local_rank, gstream.cuda_stream)
    env = {"world": world, "rank": rank, "local_rank": local_rank, "dist": dist, "dry": dry, "dev": dev, "gather": gather}

    leg = Leg(env, wl, args.path, args.seeds, args.batch_seeds, args.inflight)
    sets, synth, inflight = leg.sets, leg.synth, leg.inflight
    elapsed, results = leg.timed(args.steps, args.warmup)
    stage_ms = {k: [r["times"][k] for r, _ in results] for _, k in STAGES}
    points_done = sum(tot for _, tot in results)
    last, _ = results[-1]
    bytes_alg = sum(r["times"]["bytes_algorithmic"] for r, _ in results) / len(results)  # this rank, per step
    gstats = leg.gather_stats()
    # untimed side measurements reported beside the value
    single = e2e = tts = setup = None
    if rank == 0 and world == 1 and not args.no_extras:
        single, e2e = leg.serial_and_e2e(args.steps)
        if sets is None:
            setup = leg.setup_and_time_to_solution()
    if wl == "c4" and not args.no_extras and sets is None:
        tts = leg.time_to_solution()   # every rank takes part (collectives)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = points_done / elapsed
        avg = {k: (sum(v) / len(v) if v else 0.0) for k, v in stage_ms.items()}
        wkey = wl if sets is None else wl + "_sets"
        roof, roof_valu = _rooflines(wkey, bytes_alg, single[2][DOM_KEY] if single else None, avg[DOM_KEY])
        line = {
            "metric": "triangulated edge-points/sec", "value": value, "unit": "edge-points/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "DRY RUN of the multi-rank control flow (ranks share a GPU, no collective): not a measurement" if dry else
                    "synthetic" if wl != "c3real" else "real dtu006 edge maps + synthetic cameras and seeds", "steps_in_flight": inflight,
            "config": {
                "workload": leg.describe(), "workload_key": wkey,
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
            "roofline": roof,
        }
        if roof_valu:
            line["roofline_valu"] = roof_valu
        if gstats:
            line.update({"gather_ms": gstats["gather_ms"], "gather_bytes": gstats["gather_bytes"],
                         "overlap_frac": gstats["overlap_frac"], "gather": gstats})
            line["gather"]["mode"] = getattr(gather, "mode", None)          # "sendrecv" (default) or the "bcast" fallback
            line["gather"]["preflight"] = getattr(gather, "selftest", None)  # what the pre-flight check of the exchange found
        if tts is not None:
            line["time_to_solution_s"] = tts[0]
            line["time_to_solution"] = {"seconds": tts[0], "edge_points": tts[1], "steps": leg.n_pass_steps(),
                                        "what": "ONE pass over all %d seeds (every batch once, the last one partial), gathers "
                                                "included, contexts warm, %d steps in flight; max over ranks" % (leg.n_total, inflight)}
        if world > 1:
            # the driver derives scaling efficiency itself from ITS N = 1 run (whose line carries `scaling_base`: this
            # workload on one GPU); this is only the builder's last committed 1-GPU figure of the same workload
            for tag in ("r04", "r03"):
                ref = os.path.join(ROOT, "profiles", "%s_final_%s.json" % (tag, wkey))
                if os.path.exists(ref):
                    try:
                        r1 = json.load(open(ref))
                        line["same_workload_on_1_gpu"] = {"value": r1["value"], "ms_per_step": r1["ms_per_step"],
                                                          "time_to_solution_s": r1.get("time_to_solution_s"),
                                                          "source": "profiles/%s_final_%s.json (python bench.py --workload %s; one "
                                                                    "GPU takes the same seeds in steps of %d)" % (tag, wkey, wkey, C4_BATCH)}
                        break
                    except Exception:
                        pass
        if single is not None:
            line["ms_per_step_one_at_a_time"] = single[0] * 1e3
            line["stage_ms_one_at_a_time"] = {n: round(single[2][k], 4) for n, k in STAGES}
            line["value_one_step_at_a_time"] = single[1] / single[0]
            line["end_to_end"] = {"ms_per_step": e2e[0] * 1e3, "value": e2e[1] / e2e[0], "spread": getattr(leg, "e2e_spread", None),
                                  "what": "MEDIAN of the calls (each allocates and first-touches fresh caller-owned arrays: the host side of the copy varies "
                                          "with the OS; spread beside it); one step at a time incl. the D2H copy of the edge-point cloud into "
                                          "caller-owned host arrays (eg3d_match_resident / eg3d_match_polyline_sets, device_only=0), timed at the C ABI; "
                                          "a context that has served a host call before cuts the next ones into 3 sub-batches on its internal lanes "
                                          "(eg3d_set_pipelining default), so most of the copy runs behind the later sub-batches' kernels; a context's FIRST "
                                          "host call runs uncut (time_to_solution_one_shot.first_call_ms)"}
            line["ms_slowest_chain"] = max(r["times"].get("ms_slowest_chain", 0.0) for r, _ in results)
            line["ms_slowest_chain_what"] = ("the longest time ONE chain held its wavefront in k3b_expand (device clock): no expand launch — "
                                             "hence no lone call, however it is cut or overlapped — is shorter than its slowest chain plus the stages before it")
        if setup is None:
            line["setup_ms"] = leg.setup_ms
        else:
            line["setup_ms"] = {"eg3d_create": setup["create_ms"], "eg3d_create_again": setup["create_again_ms"],
                                "eg3d_upload_seeds": setup["upload_seeds_ms"], "first_context_of_the_process": leg.setup_ms}
            line["time_to_solution_one_shot"] = setup
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
                b, e = leg.step_range(0)
                unit = "seeds"
            # Protocol of BASELINE.md section 2: 1 warm-up + 5 timed runs, median — on a BOUNDED sample (the first third of
            # the step's seeds: ~5 s of one core per run on C3'), so that the leg stays at ~30 s of CPU work. The whole
            # step is run once more on ALL host cores: that run is the parity reference of the full step (the oracle's
            # output does not depend on the thread count: tests/test_cpu_parity.py) and the "same box's host cores" figure.
            k = args.cpu_seeds or (128 if wl == "c4" else max(1, (e - b + 2) // 3))
            ce = e if not k else min(e, b + k)
            runs = args.cpu_runs or 5
            cpu_run(b, ce, 1)  # warm-up (same sample)
            secs, pts = [], 0
            for _ in range(runs):
                r1 = cpu_run(b, ce, 1)
                secs.append(r1["stats"]["seconds"])
                pts = r1["n_points"]
            med = statistics.median(secs)
            ncores, cores_how = _usable_cores()
            ae = e if wl != "c4" else min(e, b + 1024)  # (a whole C4 step is ~1000 core-seconds: the all-core run takes its first 1024 seeds)
            cpu_run(b, min(ae, b + max(1, (ae - b) // 8)), ncores)  # warm-up of the thread team
            r = cpu_run(b, ae, ncores)
            all_value = r["n_points"] / r["stats"]["seconds"]
            line["cpu_baseline"] = {
                "value": pts / med, "unit": "edge-points/s", "cores": 1, "kind": "port",
                "sample": "%s of step 0 (%d %s, %d edge-points), oracle g++ -O3, 1 thread, 1 warm-up + %d timed runs, median "
                          "(%.2f s; runs: %s; spread %.1f %%); scene/grid construction excluded"
                          % ("all" if ce == e else "first %d %s" % (ce - b, unit), ce - b, unit, pts, len(secs), med,
                             ", ".join("%.2f" % s for s in secs), 100.0 * (max(secs) - min(secs)) / med),
                "all_cores": {"value": all_value, "cores": ncores, "cores_found_by": cores_how, "seconds": r["stats"]["seconds"],
                              "parallel_efficiency": all_value / ((pts / med) * ncores),
                              "sample": "%s (%d %s, %d edge-points), one run after a warm-up of the thread team"
                                        % ("the whole step" if ae == e else "the step's first seeds", ae - b, unit, r["n_points"])},
                "oracle_algorithmic_bytes": int(r["stats"]["bytes_algorithmic"]),
            }
            ce = ae  # (the parity check below covers what the all-core run covered: r is that run, over [b, ae))
            pts_par = r["n_points"]
            line["speedup_vs_cpu_1thread"] = value / (pts / med)
            if single is not None:
                line["speedup_vs_cpu_1thread_one_step_at_a_time"] = (single[1] / single[0]) / (pts / med)
            # "CPU-ref parity err" (the second half of BASELINE.json's metric): the GPU output of the same
            # seeds / sets copied to the host and compared with the oracle's — relative error of the 3-D
            # coordinates (north star: <= 1e-4) and exactness of every id / view list and of the order
            if sets is not None:
                gfull = leg.workers[0].ctx.match_polyline_sets(sets[0], sets[1], sets[2], b, ce)
            else:
                gfull = leg.workers[0].ctx.match_resident(b, ce)
            rep = compare_edgepoints(r, gfull, rel_tol=1e-4)
            line["parity"] = {"vs": "oracle (CPU restatement, same DLT form; parity unpinned, see DESIGN.md 3)", "points_compared": int(pts_par),
                              "max_rel_err_X": rep.get("max_rel_X"), "X_bit_exact": rep.get("bitexact_X"),
                              "ids_views_order_exact": bool(rep["ok"]), "obs_xy_bit_exact": rep.get("bitexact_xy"),
                              "tolerance": 1e-4}
    leg.close()
    # ---- sub-measurements of the default single-GPU run: the scaling workload's N = 1 point, and config 5
    if rank == 0 and world == 1 and wl == "c3" and sets is None and not (args.no_extras or args.no_sublines) and not args.seeds:
        try:
            l4 = Leg(env, "c4", inflight=args.inflight)
            el4, rs4 = l4.timed(3, 0)
            tts4 = l4.time_to_solution()
            k3b4 = sum(r["times"][DOM_KEY] for r, _ in rs4) / len(rs4)
            line["scaling_base"] = {
                "setup_ms": l4.setup_ms,
                "workload": l4.describe(), "workload_key": "c4", "n_gpus": 1, "steps": 3, "steps_in_flight": l4.inflight,
                "value": sum(t for _, t in rs4) / el4, "unit": "edge-points/s", "ms_per_step": el4 * 1e3 / 3,
                "edge_points_per_step": sum(t for _, t in rs4) / 3, "k3b_expand_ms_in_flight": k3b4,
                "time_to_solution_s": tts4[0], "time_to_solution_edge_points": tts4[1], "time_to_solution_steps": l4.n_pass_steps(),
                "what": "the N = 1 point of the multi-GPU curve (`bench.py --gpus N` measures this workload in steps of "
                        "%d x N seeds): 3 timed steps of %d seeds after one warm-up step per context, then ONE pass over all "
                        "%d seeds" % (C4_RANK_SHARE, C4_BATCH, l4.n_total)}
            l4.close()
        except Exception as ex:  # the headline line survives a failing side measurement, and says so
            line["scaling_base"] = {"error": repr(ex)}
        try:
            line["c5"] = _c5_subline(local_rank)
        except Exception as ex:
            line["c5"] = {"error": repr(ex)}
    if rank == 0:
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
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
