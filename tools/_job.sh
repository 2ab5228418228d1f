#!/bin/bash
mkdir -p gpurun_out
python tools/cpu_scaling.py 2090 --json gpurun_out/r05_cpu_scaling.json
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_arith.py 2>&1 | tail -6
