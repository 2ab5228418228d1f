#!/bin/bash
# usage (GPU box): tools/kt_stats.sh <tag> "<bench args>"  -> gpurun_out/kt_<tag>.txt : rocprofv3 --kernel-trace --stats summary (per-kernel calls / avg / min / max)
tag=$1; args=$2
out=$PWD/gpurun_out; mkdir -p $out/tmp; export TMPDIR=$out/tmp
rm -rf $out/kt_$tag
timeout -k 5 900 rocprofv3 --kernel-trace --stats -d $out/kt_$tag -o x -- python bench.py $args > $out/kt_$tag.out 2> $out/kt_$tag.err
echo "rc=$?"
python - <<PY
import sqlite3,glob,collections
rows=collections.defaultdict(list)
for db in glob.glob("$out/kt_$tag/**/*.db", recursive=True):
    c=sqlite3.connect(db)
    tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if not kd or not ks: continue
    for name,s,e in c.execute("select k.kernel_name, d.start, d.end from %s d join %s k on d.kernel_id=k.id" % (kd[0], ks[0])):
        n=name.split("(")[0].replace("void ","").replace("eg3d::","")
        rows[n[:44]].append((e-s)/1000.0)
tot=sum(sum(v) for v in rows.values())
with open("$out/kt_$tag.txt","w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python bench.py $args\n%-44s %6s %12s %10s %10s %10s %6s\n"%("kernel","calls","total_us","avg_us","min_us","max_us","pct"))
    for k,v in sorted(rows.items(), key=lambda kv:-sum(kv[1])):
        f.write("%-44s %6d %12.1f %10.2f %10.2f %10.2f %5.2f%%\n"%(k,len(v),sum(v),sum(v)/len(v),min(v),max(v),100*sum(v)/tot))
print(open("$out/kt_$tag.txt").read())
PY
tail -1 $out/kt_$tag.out > $out/kt_${tag}_bench_line.json
rm -rf $out/kt_$tag $out/tmp/*
