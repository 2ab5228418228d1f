#!/bin/bash
# GPU box: parity tests, then one-step-at-a-time timings of C2 / C3' / a C4 batch
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for c in 2 3; do INFLIGHT=1 tools/quick_bench.sh $c 6; done
python bench.py --workload c4 --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', round(d['value']), round(d['ms_per_step'],1), d['stage_ms'])"
