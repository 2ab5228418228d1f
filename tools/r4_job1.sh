#!/bin/bash
# round 4, GPU job 1: wide K3b build (2 waves/SIMD, Gauss-Newton rows kept) vs standard on C4; standard on C3'; parity of both
O=gpurun_out/r4_job1; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q -k "parity or fuzz or edge" > $O/pytest_std.log 2>&1; echo "rc=$?" >> $O/pytest_std.log )
( EG3D_K3B_WIDE=1 timeout 900 python -m pytest tests -m gpu -x -q -k "parity or fuzz" > $O/pytest_wide.log 2>&1; echo "rc=$?" >> $O/pytest_wide.log )
timeout 300 python bench.py --no-cpu-baseline > $O/c3_std.json 2> $O/c3_std.err
EG3D_K3B_WIDE=1 timeout 300 python bench.py --no-cpu-baseline > $O/c3_wide.json 2> $O/c3_wide.err
EG3D_K3B_WIDE=0 timeout 600 python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_std.json 2> $O/c4_std.err
EG3D_K3B_WIDE=1 timeout 600 python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_wide.json 2> $O/c4_wide.err
tail -3 $O/pytest_std.log $O/pytest_wide.log
for f in c3_std c3_wide c4_std c4_wide; do python - $O/$f.json <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1], "ms/step %.2f value %.3g serial %.2f k3b_excl %s stage %s" % (d["ms_per_step"], d["value"], d.get("ms_per_step_one_at_a_time",0), d["roofline"].get("exclusive",{}).get("kernel_ms_per_step"), d.get("stage_ms_one_at_a_time")))
except Exception as e: print(sys.argv[1], "FAILED", e)
P
done
