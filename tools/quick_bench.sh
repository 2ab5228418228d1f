#!/bin/bash
# usage: tools/quick_bench.sh <config> <steps> [lib]   -> prints value, ms/step, stage ms
cfg=${1:-2}; steps=${2:-10}; lib=${3:-}
[ -n "$lib" ] && export EG3D_LIB=$PWD/$lib
python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --inflight ${INFLIGHT:-1} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg$cfg', round(d['value']), round(d['ms_per_step'],2), d['stage_ms'])"
