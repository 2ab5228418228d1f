#!/bin/bash
# round 4, GPU job 2: 4 waves/SIMD (128 VGPRs, CoopLds shrunk to 8 LDS units) vs the default; window of 32 requests alone
O=gpurun_out/r4_job2; mkdir -p $O
export TMPDIR=/tmp
V=$PWD/edgegraph3d_amd/variants
( EG3D_TEST_LIB_6X4=$V/libeg3d_w4.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "dlt6x4" > $O/pytest_w4.log 2>&1; echo "rc=$?" >> $O/pytest_w4.log )
for v in default w4 w3r32; do
  lib=$PWD/edgegraph3d_amd/libeg3d.so; [ $v != default ] && lib=$V/libeg3d_$v.so
  EG3D_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-sublines > $O/c3_$v.json 2> $O/c3_$v.err
  EG3D_LIB=$lib timeout 300 python bench.py --workload c2 --no-cpu-baseline > $O/c2_$v.json 2> $O/c2_$v.err
done
for v in default w4; do
  lib=$PWD/edgegraph3d_amd/libeg3d.so; [ $v != default ] && lib=$V/libeg3d_$v.so
  EG3D_LIB=$lib timeout 600 python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu-baseline > $O/c4_$v.json 2> $O/c4_$v.err
done
tail -n 3 $O/pytest_w4.log
for f in $O/c*.json; do python - $f <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1], "ms/step %.2f value %.3g serial %.2f k3b_excl %s tts %s" % (d["ms_per_step"], d["value"], d.get("ms_per_step_one_at_a_time",0), d["roofline"].get("kernel_ms_per_step"), d.get("time_to_solution_s")))
except Exception as e: print(sys.argv[1], "FAILED", e)
P
done
