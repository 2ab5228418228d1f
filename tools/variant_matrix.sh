#!/bin/bash
# usage (GPU box): tools/variant_matrix.sh "<variant names>" [configs="2 3"]  -> gpurun_out/matrix.txt
# A/B of experimental builds (tools/build_variant.sh): one-step-at-a-time bench + K3b HBM traffic (separate PMC passes).
out=$PWD/gpurun_out; mkdir -p $out/tmp; export TMPDIR=$out/tmp
cfgs=${2:-"2 3"}
: > $out/matrix.txt
for v in $1; do
  lib=edgegraph3d_amd/variants/libeg3d_$v.so
  for c in $cfgs; do
    steps=10; [ "$c" != 2 ] && steps=4
    INFLIGHT=1 tools/quick_bench.sh $c $steps $lib 2>&1 | tail -1 | tee -a $out/matrix.txt
  done
  for ctr in FETCH_SIZE WRITE_SIZE; do
    EG3D_LIB=$PWD/$lib tools/pmc_pass.sh ${v}_$ctr "--config 2 --steps 3 --warmup 1 --no-cpu-baseline --inflight 1" $ctr 2>&1 | grep k3b | sed "s/^/$v /" | tee -a $out/matrix.txt
  done
done
