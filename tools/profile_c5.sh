#!/bin/bash
# GPU box: rocprofv3 evidence for the config-5 kernel (k5_gn_filter): --kernel-trace --stats of
# tools/bench_gn_filter.py and separate --pmc passes -> gpurun_out/r02_c5_rocprof_summary.txt
out=$PWD/gpurun_out; mkdir -p $out/tmp; export TMPDIR=$out/tmp
sum=$out/r02_c5_rocprof_summary.txt
echo "# rocprofv3 -- python tools/bench_gn_filter.py  (1 M points, 16-view rig; see profiles/r02_final_c5.json)" > $sum
pass() { # name, rocprof flags...
  local name=$1; shift
  rm -rf $out/prof_$name
  timeout -k 5 300 rocprofv3 "$@" -d $out/prof_$name -o x -- python tools/bench_gn_filter.py > $out/prof_$name.out 2> $out/prof_$name.err
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
rm -rf $out/tmp/*
cat $sum
