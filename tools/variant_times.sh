#!/bin/bash
# usage (GPU box): tools/variant_times.sh "<variant names>"  -> one-step-at-a-time stage times on C2, C3', a 2048-seed C4 batch
for v in $1; do
  lib=edgegraph3d_amd/variants/libeg3d_$v.so
  for c in 2 3; do INFLIGHT=1 tools/quick_bench.sh $c 6 $lib 2>&1 | tail -1; done
  EG3D_LIB=$PWD/$lib python bench.py --workload c4 --batch-seeds 2048 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c4/2048', round(d['value']), round(d['ms_per_step'],1), d['stage_ms'])"
done
