#!/usr/bin/env python3
"""Turn rocprofv3 rocpd SQLite outputs (gpurun_out/prof_*/…_results.db) into the text/JSON
summaries committed under profiles/.

  python tools/profile_summary.py --tag r01 --kt gpurun_out/prof_kt/r01_results.db \
      --fetch gpurun_out/prof_fetch/r01_results.db --write gpurun_out/prof_write/r01_results.db \
      --sq gpurun_out/prof_sq/r01_results.db

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are collected in separate --pmc passes, are reported in KiB, and on gfx950
FETCH_SIZE reads half of the bytes of wide coalesced streams (128 B requests tallied as 64 B), so
both the raw value and the x2-corrected upper bound are listed.
"""
import argparse
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    if re.match(r"(?:void )?(?:eg3d::)?k3b_expand_t<", name):
        # the instantiations of the expand kernel <waves per SIMD, Gauss-Newton chunks kept, scene class>: a context runs
        # exactly one of them, so they share the row "k3b_expand"
        return "k3b_expand"
    m = re.match(r"(?:void )?(?:eg3d::)?([A-Za-z0-9_]+)(<[a-z]+>)?", name)
    if "rocprim" in name:
        return "rocprim::scan(init)" if "init_lookback" in name else "rocprim::scan"
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def kernel_stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration from kernels").fetchall()
    agg = defaultdict(list)
    for n, d in rows:
        agg[short(n)].append(d)
    total = sum(sum(v) for v in agg.values())
    out = []
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        out.append({"kernel": k, "calls": len(v), "total_us": sum(v) / 1e3, "avg_us": sum(v) / len(v) / 1e3,
                    "min_us": min(v) / 1e3, "max_us": max(v) / 1e3, "pct": 100.0 * sum(v) / total})
    return out


def counter_stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = defaultdict(lambda: defaultdict(list))
    for n, c, v in rows:
        agg[short(n)][c].append(v)
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in agg.items()}


def counter_sums(path):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = defaultdict(lambda: defaultdict(float))
    for n, c, v in rows:
        agg[short(n)][c] += v
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--kt")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq", nargs="*")
    ap.add_argument("--cmd", default="python bench.py --steps 20 --warmup 3 --no-cpu-baseline")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--workload", default=None, help="key under which the traffic is stored in pmc_traffic.json (c2, c3, c4, ...)")
    ap.add_argument("--pmc-steps", type=int, default=0,
                    help="hot-path passes executed by the PMC runs (max(warmup, inflight) + steps with --no-extras): "
                         "traffic per step = sum over the launches / this")
    ap.add_argument("--pmc-cmd", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    lines = []
    if a.kt:
        ks = kernel_stats(a.kt)
        lines.append("# rocprofv3 --kernel-trace --stats -- %s" % a.cmd)
        lines.append("%-28s %6s %12s %12s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
        for k in ks:
            lines.append("%-28s %6d %12.1f %12.2f %10.2f %10.2f %6.2f%%" % (k["kernel"], k["calls"], k["total_us"],
                                                                          k["avg_us"], k["min_us"], k["max_us"], k["pct"]))
        json.dump(ks, open(os.path.join(a.out, "%s_kernel_stats.json" % a.tag), "w"), indent=1)
    traffic = {}
    for label, path in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not path:
            continue
        cs = counter_stats(path)
        lines.append("")
        lines.append("# rocprofv3 --pmc %s --kernel-trace (own pass)  [KiB per launch, mean]" % label)
        for k, d in sorted(cs.items(), key=lambda kv: -kv[1].get(label, (0, 0))[0]):
            if label in d and d[label][0] > 0:
                kib, n = d[label]
                lines.append("%-28s %12.1f KiB  (%d launches)" % (k, kib, n))
                traffic.setdefault(k, {})[label] = kib * 1024.0
    if traffic:
        out = {}
        for k, d in traffic.items():
            f, w = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)
            out[k] = {"fetch_bytes_raw": f, "write_bytes_raw": w, "hbm_bytes_per_launch": 2.0 * f + w,
                      "note": "FETCH_SIZE x2 (gfx950 wide-read correction, upper bound) + WRITE_SIZE"}
        if a.workload and a.pmc_steps and a.fetch and a.write:
            fs, ws = counter_sums(a.fetch), counter_sums(a.write)
            for k in out:
                tot = 2.0 * fs.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 + ws.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
                out[k]["hbm_bytes_per_step"] = tot / a.pmc_steps
                # k3b_expand / k4_emit run once per steady-state step; the first passes of a context repeat the launch
                # while its capacities and staging area grow (C4: every pass of a 3-pass run): per step = one launch
                n_l = (counter_stats(a.fetch).get(k, {}).get("FETCH_SIZE") or (0, 0))[1]
                if k in ("k3b_expand", "k4_emit") and n_l > a.pmc_steps:
                    out[k]["hbm_bytes_per_step"] = out[k]["hbm_bytes_per_launch"]
                    out[k]["note_step"] = ("%d launches in %d passes (a context's first passes repeat the launch while its "
                                           "buffers grow): a steady-state step is ONE launch, per step = the per-launch mean"
                                           % (n_l, a.pmc_steps))
            path = os.path.join(a.out, "pmc_traffic.json")
            allw = json.load(open(path)) if os.path.exists(path) else {}
            if not isinstance(allw, dict) or any(isinstance(v, dict) and "hbm_bytes_per_launch" in v for v in allw.values()):
                allw = {}  # round-1 layout (one flat workload): start over
            ent = dict(out)
            ent["provenance"] = ("profiles/%s_rocprof_summary.txt: separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE "
                                 "passes of `%s` (%d passes of the hot path each); bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B), "
                                 "summed over the kernel's launches / passes" % (a.tag, a.pmc_cmd, a.pmc_steps))
            try:
                sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
                from edgegraph3d_amd import build as _build
                ent["source_fingerprint"] = _build.device_source_fingerprint()
            except Exception:
                pass
            allw[a.workload] = ent
            json.dump(allw, open(path, "w"), indent=1)
        else:
            json.dump(out, open(os.path.join(a.out, "pmc_traffic.json"), "w"), indent=1)
        lines.append("")
        lines.append("# HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (guide's gfx950 correction; upper bound)")
        for k, d in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
            lines.append("%-28s %14.0f B" % (k, d["hbm_bytes_per_launch"]))
    for sq in (a.sq or []):
        if not os.path.exists(sq):
            continue
        cs = counter_stats(sq)
        lines.append("")
        lines.append("# rocprofv3 --pmc SQ_* --kernel-trace (own pass) [mean per launch]")
        for k, d in cs.items():
            if not k.startswith("k"):
                continue
            lines.append("%-28s %s" % (k, "  ".join("%s=%.3g" % (c, v[0]) for c, v in sorted(d.items()))))
    # derived: fraction of the 64 lanes active in the VALU instructions of the dominant kernels
    merged = defaultdict(dict)
    for sq in (a.sq or []):
        if os.path.exists(sq):
            for k, d in counter_stats(sq).items():
                merged[k].update({c: v[0] for c, v in d.items()})
    # vector-ALU figures of the expand kernel go into pmc_traffic.json beside its traffic (bench.py: roofline_valu)
    if a.workload and merged:
        path = os.path.join(a.out, "pmc_traffic.json")
        allw = json.load(open(path)) if os.path.exists(path) else {}
        ent = allw.setdefault(a.workload, {})
        kt_avg = {}
        if a.sq and os.path.exists(a.sq[0]):
            for k in kernel_stats(a.sq[0]):
                kt_avg[k["kernel"]] = k["avg_us"] / 1e3
        for k, d in merged.items():
            if k.startswith("k3b_expand") and d.get("SQ_THREAD_CYCLES_VALU"):
                tgt = ent.setdefault(k, {})
                tgt["valu"] = {c: d.get(c) for c in ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                                     "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU") if d.get(c) is not None}
                tgt["valu"]["active_lane_frac"] = d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])
                tgt["valu"]["kernel_ms"] = kt_avg.get(k)
                tgt["valu"]["kernel"] = k
                ent["provenance_valu"] = ("profiles/%s_rocprof_summary.txt: rocprofv3 --kernel-trace --pmc SQ_* passes of `%s`; means per "
                                          "launch of %s" % (a.tag, a.pmc_cmd, k))
        json.dump(allw, open(path, "w"), indent=1)
    der = []
    for k, d in merged.items():
        if d.get("SQ_ACTIVE_INST_VALU") and d.get("SQ_THREAD_CYCLES_VALU") and k.startswith("k3"):
            der.append("%-28s active-lane fraction of VALU instructions = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.3f"
                       % (k, d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])))
    if der:
        lines.append("")
        lines.append("# derived")
        lines.extend(der)
    open(os.path.join(a.out, "%s_rocprof_summary.txt" % a.tag), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
