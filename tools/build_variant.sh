#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> edgegraph3d_amd/variants/libeg3d_<name>.so
# (experimental builds for A/B timing on the GPU box: tools/quick_bench.sh <cfg> <steps> edgegraph3d_amd/variants/libeg3d_<name>.so)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p edgegraph3d_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fno-slp-vectorize -fno-vectorize -mllvm -disable-machine-licm -mllvm -disable-lsr -mllvm -enable-pre=false -mllvm -enable-load-pre=false -mllvm -enable-misched=false -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -Wno-unused-function -I include -I edgegraph3d_amd/csrc -I edgegraph3d_amd/host \
  "$@" -o edgegraph3d_amd/variants/libeg3d_$name.so edgegraph3d_amd/csrc/eg3d_api.hip edgegraph3d_amd/csrc/eg3d_kernels.hip \
  edgegraph3d_amd/host/grid_build.cpp
