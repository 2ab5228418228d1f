#!/bin/bash
# GPU box, repo root: the whole validation + measurement job of a round: GPU test suite, smoke, the four rocprofv3 profile sets
# (tools/profile_r03.sh), every bench line DESIGN.md quotes (tools/r3_final_numbers.sh), the K3a engine statistics of the timing build
# (build it first: tools/build_variant.sh timing -DEG3D_SECTION_TIMING). Results under gpurun_out/.
cd /root/repo
mkdir -p gpurun_out
(timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/final_pytest.log
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > gpurun_out/final_smoke.log
for wl in c3 c2 c3real c4 c2_sets c3_sets; do bash tools/profile_r03.sh $wl > gpurun_out/profile_$wl.log 2>&1; done
bash tools/r3_final_numbers.sh > gpurun_out/final_numbers.log 2>&1
export EG3D_LIB=$PWD/edgegraph3d_amd/variants/libeg3d_timing.so
(timeout 600 python tools/k3a_stats.py 3; timeout 600 python tools/k3a_stats.py 2) > gpurun_out/r03_k3a_engine_stats.txt 2>&1
cat gpurun_out/final_pytest.log gpurun_out/final_smoke.log gpurun_out/final_numbers.log
