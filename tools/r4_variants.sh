#!/bin/bash
# usage (GPU box): tools/r4_variants.sh <out dir> <workloads "c3 c2 c4"> <variant names...>   ("default" = the product library)
O=gpurun_out/$1; mkdir -p $O; shift
WLS=$1; shift
export TMPDIR=/tmp
V=$PWD/edgegraph3d_amd/variants
for v in "$@"; do
  lib=$PWD/edgegraph3d_amd/libeg3d.so; [ $v != default ] && lib=$V/libeg3d_$v.so
  for wl in $WLS; do
    args="--workload $wl --no-cpu-baseline --no-sublines"; [ $wl = c4 ] && args="--workload c4 --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
    EG3D_LIB=$lib timeout 600 python bench.py $args > $O/${wl}_$v.json 2> $O/${wl}_$v.err
    python - $O/${wl}_$v.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1], "ms/step %.2f value %.4g serial %.2f k3b %s stages %s" % (d["ms_per_step"], d["value"], d.get("ms_per_step_one_at_a_time",0), d["roofline"].get("kernel_ms_per_step"), d.get("stage_ms_one_at_a_time")))
except Exception as e: print(sys.argv[1], "FAILED", e)
P
  done
done
