#!/bin/bash
# GPU box: default bench (C3'), the 1-rank RCCL gather path inside the torch process, and a short C4 run
mkdir -p gpurun_out
(time python bench.py --steps 20 --warmup 5) > gpurun_out/b_c3.json 2> gpurun_out/b_c3.err; tail -3 gpurun_out/b_c3.err
python bench.py --steps 5 --warmup 2 --force-gather --no-cpu-baseline > gpurun_out/b_c3_gather.json 2> gpurun_out/b_c3_gather.err; tail -3 gpurun_out/b_c3_gather.err
(time python bench.py --workload c4 --steps 3 --warmup 1 --inflight 2) > gpurun_out/b_c4.json 2> gpurun_out/b_c4.err; tail -3 gpurun_out/b_c4.err
cat gpurun_out/b_c3.json gpurun_out/b_c3_gather.json gpurun_out/b_c4.json
