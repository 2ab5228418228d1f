#!/bin/bash
# usage (GPU box): tools/traffic.sh <tag> "<bench args>"  -> gpurun_out/traffic_<tag>.txt (FETCH_SIZE / WRITE_SIZE per kernel, separate passes)
tag=$1; args=$2
: > gpurun_out/traffic_$tag.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  tools/pmc_pass.sh ${tag}_$ctr "$args" $ctr 2>&1 | grep -E "k3|k2|k1" | tee -a gpurun_out/traffic_$tag.txt
done
