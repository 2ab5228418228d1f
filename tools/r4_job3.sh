#!/bin/bash
# round 4, GPU job 3: the whole GPU suite on the new default build (4 waves/SIMD), smoke, the driver's bench command
O=gpurun_out/r4_job3; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log )
( python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 )
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
tail -n 15 $O/pytest.log; tail -n 2 $O/smoke.log; cat $O/bench_default.time; tail -n 3 $O/bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r4_job3/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_one_at_a_time","value_one_step_at_a_time")})
print("roofline", {k:d["roofline"].get(k) for k in ("achieved","frac","kernel_ms_per_step","traffic","measured_with")}, d["roofline"].get("in_flight"))
print("valu", d.get("roofline_valu"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("speedup_vs_cpu_1thread"), d.get("parity"))
print("scaling_base", d.get("scaling_base"))
print("c5", d.get("c5"))
P
