#!/usr/bin/env python3
"""How does the CPU oracle (the `cpu_baseline` of bench.py) scale over the host's cores, and what does the box really
give this process? Prints the affinity / cgroup limits and edge-points/s of the first N seeds of C3' at 1, 2, 4, ...
threads. usage: tools/cpu_scaling.py [n_seeds=2090] [--json out]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from edgegraph3d_amd import host
from oracle import binding as ob

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2090
info = {"os_cpu_count": os.cpu_count(), "sched_getaffinity": len(os.sched_getaffinity(0))}
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        info[f] = open(f).read().strip()
    except OSError:
        pass
try:
    info["loadavg"] = open("/proc/loadavg").read().strip()
except OSError:
    pass
print(info, flush=True)
s = host.Synth(3)
o = ob.Oracle(s.scene)
o.match(s.seeds, 0, 200, 1)
rows = []
th = 1
base = None
while th <= (os.cpu_count() or 1):
    o.match(s.seeds, 0, min(n, 400), th)  # team warm-up
    t0 = time.time()
    r = o.match(s.seeds, 0, n, th)
    wall = time.time() - t0
    v = r["n_points"] / r["stats"]["seconds"]
    base = base or v
    rows.append({"threads": th, "seconds": r["stats"]["seconds"], "wall_incl_packing": wall, "edge_points_per_s": v, "speedup": v / base,
                 "efficiency": v / base / th})
    print("threads %3d  %.3f s (%.3f s with output packing)  %.0f edge-points/s  x%.1f  efficiency %.2f" % (th, r["stats"]["seconds"], wall, v, v / base, v / base / th), flush=True)
    th *= 2
if "--json" in sys.argv:
    json.dump({"host": info, "seeds": n, "rows": rows}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
