#!/bin/bash
O=gpurun_out/r4_gather; mkdir -p $O
python tests/rccl_single_rank_check.py > $O/single.log 2>&1; tail -n 3 $O/single.log
python bench.py --workload c2 --steps 5 --warmup 2 --force-gather --no-cpu-baseline > $O/force_gather.json 2> $O/force_gather.err; python -c "
import json; d=json.load(open('$O/force_gather.json')); print('force-gather', d['value'], d.get('gather'))"
bash tests/bench_dryrun_check.sh 2>&1 | tail -n 2
