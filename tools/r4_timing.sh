#!/bin/bash
# section clocks + GN statistics of the timing build on C3' (3) and C2 (2)
O=gpurun_out/r4_timing; mkdir -p $O
export EG3D_LIB=$PWD/edgegraph3d_amd/variants/libeg3d_timing.so
for c in 3 2; do
  timeout 300 python tools/section_timing.py $c > $O/sections_c$c.txt 2>&1
  timeout 300 python tools/gn_stats.py $c > $O/gn_c$c.txt 2>&1
done
cat $O/sections_c3.txt $O/gn_c3.txt
