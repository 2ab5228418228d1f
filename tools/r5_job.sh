#!/bin/bash
mkdir -p gpurun_out
export EG3D_K3B_ENGINE=1
for l in 2 4 8 16; do echo "w2 lanes=$l"; EG3D_K3C_LANES=$l INFLIGHT=1 timeout 600 tools/quick_bench.sh 3 4; done
echo "w2 lanes=8 in flight 4"; EG3D_K3C_LANES=8 INFLIGHT=4 timeout 600 tools/quick_bench.sh 3 8
echo "c2 lanes=4"; EG3D_K3C_LANES=4 INFLIGHT=1 timeout 600 tools/quick_bench.sh 2 6
