#!/bin/bash
# usage: tools/section_light.sh build "<sections>"   (here: cross-compiles one light timing variant per section)
#        tools/section_light.sh run "<sections>" [config]   (GPU box: one line per section -> stdout)
# Sections: edgegraph3d_amd/csrc/eg3d_dev_expand.h Chain::tsec.
mode=$1; secs=${2:-"0 1 2 3 4 8 10 12 13"}; cfg=${3:-3}
cd "$(dirname "$0")/.."
case $mode in
  build) for k in $secs; do tools/build_variant.sh sec$k -DEG3D_SECTION_TIMING -DEG3D_ONE_SECTION=$k > /tmp/sec$k.log 2>&1 || { echo "sec$k FAILED"; tail -n 5 /tmp/sec$k.log; }; done;;
  run) for k in $secs; do EG3D_LIB=$PWD/edgegraph3d_amd/variants/libeg3d_sec$k.so timeout 300 python tools/section_light.py $k $cfg 2>&1 | tail -n 1; done;;
esac
