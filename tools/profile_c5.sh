#!/bin/bash
# GPU box: rocprofv3 evidence for the config-5 kernel (k5_gn_filter): --kernel-trace --stats of
# tools/bench_gn_filter.py and separate --pmc passes -> gpurun_out/<tag>_c5_rocprof_summary.txt   usage: profile_c5.sh [tag=r06]
tag=${1:-r06}
out=$PWD/gpurun_out; mkdir -p $out/tmp; export TMPDIR=$out/tmp
sum=$out/${tag}_c5_rocprof_summary.txt
echo "# rocprofv3 -- python tools/bench_gn_filter.py --no-cpu  (1 M points, 16-view rig; iteration statistics: profiles/r06_c5_iterations.json)" > $sum
pass() { # name, rocprof flags...
  local name=$1; shift
  rm -rf $out/prof_$name
  timeout -k 5 300 rocprofv3 "$@" -d $out/prof_$name -o x -- python tools/bench_gn_filter.py --no-cpu > $out/prof_$name.out 2> $out/prof_$name.err
  python - "$out/prof_$name" "$name" "$*" >> $sum <<'PY'
import sqlite3, glob, sys, collections
d, name, flags = sys.argv[1:4]
print("\n# pass %s: rocprofv3 %s" % (name, flags))
for db in glob.glob(d + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    if "pmc" in name:
        agg = collections.defaultdict(list)
        for n, cn, v in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if "k5_gn_filter" in n:
                agg[cn].append(v)
        for cn, v in sorted(agg.items()):
            print("k5_gn_filter  %-28s mean per launch %.6g  (%d launches)" % (cn, sum(v) / len(v), len(v)))
    else:
        rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc")) if "kernels" in tabs else []
        for r in rows[:6]:
            print("%-40s calls %4d  total %.1f us  avg %.2f us  min %.2f  max %.2f" % (r[0][:40], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
PY
  rm -rf $out/prof_$name
}
pass kt --kernel-trace --stats
pass pmc_fetch --kernel-trace --pmc FETCH_SIZE
pass pmc_write --kernel-trace --pmc WRITE_SIZE
pass pmc_sq --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
pass pmc_lanes --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
python - $sum <<'PY'
import re, sys
t = open(sys.argv[1]).read()
def last(name):
    m = re.findall(r"k5_gn_filter\s+%s\s+mean per launch ([0-9.e+]+)" % name, t)
    return float(m[-1]) if m else None
th, act = last("SQ_THREAD_CYCLES_VALU"), last("SQ_ACTIVE_INST_VALU")
if th and act:
    open(sys.argv[1], "a").write("\n# active-lane fraction of k5_gn_filter = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.4f\n" % (th / (64.0 * act)))
import json
m = re.search(r"k5_gn_filter.*?calls\s+\d+\s+total [0-9.]+ us\s+avg ([0-9.]+) us", t)
json.dump({"kernel": "k5_gn_filter", "kernel_ms": float(m.group(1)) / 1e3 if m else None, "SQ_THREAD_CYCLES_VALU": th,
           "SQ_ACTIVE_INST_VALU": act, "SQ_INSTS_VALU": last("SQ_INSTS_VALU"), "SQ_WAVES": last("SQ_WAVES"), "SQ_WAIT_ANY": last("SQ_WAIT_ANY"),
           "SQ_WAVE_CYCLES": last("SQ_WAVE_CYCLES"), "FETCH_SIZE": last("FETCH_SIZE"), "WRITE_SIZE": last("WRITE_SIZE"),
           "active_lane_frac": th / (64.0 * act) if th and act else None,
           "source": "tools/profile_c5.sh -> " + sys.argv[1].split("/")[-1]}, open(sys.argv[1].replace("_rocprof_summary.txt", "_valu.json"), "w"), indent=1)
PY
rm -rf $out/tmp/*
cat $sum
