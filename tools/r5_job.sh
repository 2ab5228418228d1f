#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; tail -c 600 gpurun_out/r5_bench_default.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus")})
print("roofline", d.get("roofline")); print("cpu_baseline", d.get("cpu_baseline")); print("parity", d.get("parity"))
print("speedup", d.get("speedup_vs_cpu_1thread"), "one at a time", d.get("ms_per_step_one_at_a_time"))
PY
