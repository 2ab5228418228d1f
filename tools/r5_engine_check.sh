#!/bin/bash
# GPU box, round 5: the lane-per-chain engine (EG3D_K3B_ENGINE=1) against the oracle, then timings next to k3b_expand
mkdir -p gpurun_out
export EG3D_K3B_ENGINE=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15
for c in 2 3; do
  for e in 1 0; do
    echo "engine=$e"; EG3D_K3B_ENGINE=$e INFLIGHT=1 timeout 600 tools/quick_bench.sh $c 6
  done
done
