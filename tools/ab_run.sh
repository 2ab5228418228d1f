#!/bin/bash
# GPU box: A/B of library variants, one step on the GPU at a time (and 4 in flight for C3').
# usage: tools/ab_run.sh "<lib paths, '-' = the product library>" [rounds]
libs=${1:-"-"}; rounds=${2:-2}
for r in $(seq 1 $rounds); do
  for lib in $libs; do
    [ "$lib" = "-" ] && l="" || l=$lib
    INFLIGHT=1 tools/quick_bench.sh 3 8 $l
  done
done
for lib in $libs; do
  [ "$lib" = "-" ] && l="" || l=$lib
  INFLIGHT=4 tools/quick_bench.sh 3 12 $l
  INFLIGHT=1 tools/quick_bench.sh 2 10 $l
  [ -n "$l" ] && export EG3D_LIB=$PWD/$l || unset EG3D_LIB
  python bench.py --workload c4 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib c4', round(d['value']), round(d['ms_per_step'],1), d['stage_ms'])"
done
