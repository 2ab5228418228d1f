#!/bin/bash
O=gpurun_out/r4_quick2; mkdir -p $O; export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py -m gpu -x -q -k "dlt6x4" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log )
tail -n 4 $O/pytest.log
bash tools/r4_variants.sh r4_quick2 "${1:-c3 c2 c4}" default
