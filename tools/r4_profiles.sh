#!/bin/bash
# round 4: the profile sets of the final build (C3', C2, C4) + the bench lines DESIGN quotes
cd /root/repo; mkdir -p gpurun_out
for wl in c3 c2 c4; do bash tools/profile_r04.sh $wl > gpurun_out/profile_r04_$wl.log 2>&1; tail -n 25 gpurun_out/profile_r04_$wl.log; done
run() { name=$1; shift; ( "$@" > gpurun_out/r04_final_$name.json 2> gpurun_out/r04_final_$name.err ); python - gpurun_out/r04_final_$name.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], "value %.4g ms/step %.2f serial %s tts %s" % (d["value"], d["ms_per_step"], d.get("ms_per_step_one_at_a_time"), d.get("time_to_solution_s")))
except Exception as e: print(sys.argv[1], "FAILED", e)
P
}
run c3 python bench.py --gpus 1 --steps 20 --warmup 5
run c2 python bench.py --workload c2 --steps 20 --warmup 5
run c3real python bench.py --workload c3real --steps 20 --warmup 5
run c4 python bench.py --workload c4 --steps 4 --warmup 1
run sets_c2 python bench.py --workload c2 --path sets --steps 10 --warmup 3
run sets_c3 python bench.py --workload c3 --path sets --steps 4 --warmup 1
