#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace CSV of tools/timeline_probe.py into a per-call table of the path's kernels (start,
end, duration in ms from the call's first kernel; HSA queue = lane). usage: timeline_summary.py <kernel_trace.csv> [last N calls]"""
import csv
import sys

NAMES = ("k3b_expand", "k3a_orient", "k3a_follow", "k2_epipolar", "k1_seed_cand", "k4_emit")
rows = list(csv.DictReader(open(sys.argv[1])))
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def short(n):
    for k in NAMES:
        if k in n:
            return k
    return None


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows if short(r["Kernel_Name"]))
calls = [[ev[0]]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in calls[-1]) > 100e6:   # the probe sleeps 200 ms between timed calls
        calls.append([e])
    else:
        calls[-1].append(e)
for c in calls[-keep:]:
    t0 = c[0][0]
    span = (max(e[1] for e in c) - t0) / 1e6
    if span > 150:
        continue  # (warm-up calls run back to back: not one call)
    print("---- one call: first kernel to last kernel %.2f ms, %d queues" % (span, len(set(e[3] for e in c))))
    for s, e, n, q in c:
        if n in ("k3b_expand", "k3a_follow", "k3a_orient", "k4_emit"):
            print("  %-11s queue %s  %7.2f -> %7.2f  (%6.2f ms)" % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
