#!/bin/bash
# GPU box: quick A/B on C3' (one step at a time) and C2; usage: tools/ab_c3.sh "<libs, '-' = product>" [rounds]
libs=${1:-"-"}; rounds=${2:-2}
for r in $(seq 1 $rounds); do
  for lib in $libs; do [ "$lib" = "-" ] && l="" || l=$lib; INFLIGHT=1 tools/quick_bench.sh 3 8 $l; done
done
for lib in $libs; do [ "$lib" = "-" ] && l="" || l=$lib; INFLIGHT=1 tools/quick_bench.sh 2 10 $l; done
