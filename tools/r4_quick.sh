#!/bin/bash
# quick GPU check of the current default build: parity subset (6x4 form) + C3'/C2 (+C4 with arg) bench lines
O=gpurun_out/r4_quick; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "dlt6x4" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log )
tail -n 3 $O/pytest.log
bash tools/r4_variants.sh r4_quick "${1:-c3 c2}" default
