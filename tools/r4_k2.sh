#!/bin/bash
# GPU box: parity of the flat-segment K2, then A/B lines (wave-priority variants, steps in flight)
export TMPDIR=/tmp
O=gpurun_out/r4_k2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stage_a or full_path or polyline_sets" 2>&1 | tail -n 4
bash tools/r4_variants.sh r4_k2 "c3" default prio_gn0 prio_gn3
for n in 3 6 8; do
  timeout 600 python bench.py --workload c3 --no-cpu-baseline --no-sublines --inflight $n > $O/c3_inflight$n.json 2> $O/c3_inflight$n.err
  python -c "import json,sys; d=json.load(open('$O/c3_inflight$n.json')); print('inflight $n', d['ms_per_step'], d['value'])"
done
bash tools/r4_variants.sh r4_k2 "c2 c4" default prio_gn0
