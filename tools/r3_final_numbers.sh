#!/bin/bash
# GPU box: every bench line DESIGN.md quotes, with the current build -> gpurun_out/final_*.json
mkdir -p gpurun_out
run() { name=$1; shift; (time "$@") > gpurun_out/final_$name.json 2> gpurun_out/final_$name.err; tail -4 gpurun_out/final_$name.err | grep real; }
run c3 python bench.py --gpus 1 --steps 20 --warmup 5
run c2 python bench.py --workload c2 --steps 20 --warmup 5
run c3real python bench.py --workload c3real --steps 20 --warmup 5
run c4 python bench.py --workload c4 --steps 4 --warmup 1
run sets_c2 python bench.py --workload c2 --path sets --steps 10 --warmup 3
run sets_c3 python bench.py --workload c3 --path sets --steps 4 --warmup 1 --cpu-runs 1
run c5 python tools/bench_gn_filter.py
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    cb=d.get('cpu_baseline') or {}
    if 'value' not in d:
        print(f.split('final_')[1][:-5], {k: d[k] for k in ('kernel_ms','points_per_s','speedup_vs_cpu_1thread') if k in d}); continue
    print(f.split('final_')[1][:-5], 'value %.4g'%d['value'], 'ms %.3f'%d.get('ms_per_step',0), 'serial', d.get('value_one_step_at_a_time'), d.get('ms_per_step_one_at_a_time'),
          'e2e', (d.get('end_to_end') or {}).get('ms_per_step'), 'cpu', cb.get('value'), cb.get('all_cores'), 'parity', d.get('parity'))
PY
