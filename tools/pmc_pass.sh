#!/bin/bash
# usage (GPU box): tools/pmc_pass.sh <name> "<bench args>" COUNTER [COUNTER...]   -> gpurun_out/pmc_<name>.txt
# NOTE: one derived counter family per pass (FETCH_SIZE and WRITE_SIZE together exceed the hardware and rocprofv3 then hangs in its abort handler)
name=$1; shift; args=$1; shift
out=$PWD/gpurun_out; mkdir -p $out/tmp; export TMPDIR=$out/tmp
rm -rf $out/prof_$name
timeout -k 5 300 rocprofv3 --kernel-trace --pmc "$@" -d $out/prof_$name -o x -- python bench.py $args > $out/prof_$name.out 2> $out/prof_$name.err
echo "rc=$?"
python - <<PY
import sqlite3,collections,glob
for db in glob.glob("$out/prof_$name/**/*.db", recursive=True):
    c=sqlite3.connect(db)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for n,cn,v in c.execute("select kernel_name, counter_name, value from counters_collection"):
        agg[n[:40]][cn].append(v)
    with open("$out/pmc_$name.txt","w") as f:
        for k,d in agg.items():
            if "k3" in k or "k2" in k or "k1" in k:
                line=k+" "+"  ".join("%s=%.4g"%(c_,sum(v)/len(v)) for c_,v in sorted(d.items()))
                print(line); f.write(line+"\n")
PY
tail -3 $out/prof_$name.err
rm -rf $out/prof_$name $out/tmp/*  # keep gpurun_out under the 64 MiB copy-back limit
