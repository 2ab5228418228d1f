#!/bin/bash
# The host library (N2 builder, PNG reader, replay, F estimation, JSON, post steps, scene generator) built with
# AddressSanitizer + UBSan and run through its CPU tests. No GPU needed. Restores the normal library afterwards.
set -e
cd "$(dirname "$0")/.."
lib=edgegraph3d_amd/libeg3d_host.so
tmp=$(mktemp -d)
cp $lib $tmp/orig.so
trap 'cp $tmp/orig.so $lib' EXIT
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined \
    -fno-omit-frame-pointer -I include -I edgegraph3d_amd/csrc -o $lib edgegraph3d_amd/host/*.cpp -lz
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
    python -m pytest tests/test_plg_build.py tests/test_replay.py tests/test_fmatrix.py tests/test_host_io.py \
    tests/test_cpu_fuzz.py -x -q "$@" 2>&1 | tee $tmp/run.log | tail -3
echo "UBSan reports: $(grep -c 'runtime error' $tmp/run.log || true)"
