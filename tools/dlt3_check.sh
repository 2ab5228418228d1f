#!/bin/bash
# GPU box: the 6x4-DLT build against the oracle in the same mode (parity tests), then timings of both builds
export EG3D_LIB=$PWD/edgegraph3d_amd/variants/libeg3d_dlt3.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -x -q -k "not golden and not refapi and not edge_matcher and not rccl" 2>&1 | tail -4
unset EG3D_LIB
tools/variant_times.sh "dlt2 dlt3"
