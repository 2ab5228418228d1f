#!/bin/bash
# usage (GPU box, repo root): tools/final_numbers.sh <tag, e.g. r06> [workloads to profile, default "c3 c2 c4"]
# The profile sets of a round's final build (tools/profile_round.sh per workload) + the bench lines DESIGN_LOG.md quotes,
# all under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
tag=${1:-r06}; wls=${2:-"c3 c2 c4"}
mkdir -p gpurun_out
for wl in $wls; do bash tools/profile_round.sh $wl $tag > gpurun_out/profile_${tag}_$wl.log 2>&1; tail -n 25 gpurun_out/profile_${tag}_$wl.log; done
run() { name=$1; shift; ( "$@" > gpurun_out/${tag}_final_$name.json 2> gpurun_out/${tag}_final_$name.err ); python - gpurun_out/${tag}_final_$name.json <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], "value %.4g ms/step %.2f serial %s tts %s" % (d["value"], d["ms_per_step"], d.get("ms_per_step_one_at_a_time"), d.get("time_to_solution_s")))
except Exception as e: print(sys.argv[1], "FAILED", e)
P
}
run c3 python bench.py --gpus 1 --steps 20 --warmup 5
run c2 python bench.py --workload c2 --steps 20 --warmup 5
run c3real python bench.py --workload c3real --steps 20 --warmup 5
run c4 python bench.py --workload c4 --steps 4 --warmup 1
