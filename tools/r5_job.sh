#!/bin/bash
# GPU box, round 5: engine variants
mkdir -p gpurun_out
export EG3D_K3B_ENGINE=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for l in 8 16 32; do echo "w2 lanes=$l"; EG3D_K3C_LANES=$l INFLIGHT=1 timeout 600 tools/quick_bench.sh 3 4; done
for w in 3 4; do for l in 8 16; do echo "w$w lanes=$l"; EG3D_K3C_LANES=$l INFLIGHT=1 timeout 600 tools/quick_bench.sh 3 4 edgegraph3d_amd/variants/libeg3d_w$w.so; done; done
EG3D_K3C_LANES=16 EG3D_LIB=$PWD/edgegraph3d_amd/variants/libeg3d_timing.so timeout 600 python tools/k3c_stats.py 3
echo "w2 c2 lanes=8"; EG3D_K3C_LANES=8 INFLIGHT=1 timeout 600 tools/quick_bench.sh 2 6
