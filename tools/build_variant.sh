#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  -> edgegraph3d_amd/variants/libeg3d_<name>.so
# (experimental builds for A/B timing on the GPU box: tools/quick_bench.sh <cfg> <steps> edgegraph3d_amd/variants/libeg3d_<name>.so)
# The switches are edgegraph3d_amd/build.py's (one list, no hand copy); the build guard is off for variants.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p edgegraph3d_amd/variants
EG3D_NO_BUILD_GUARD=1 python - "$name" "$@" <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from edgegraph3d_amd import build
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(build.PKG, "variants", "libeg3d_%s.so" % name)
build.build_hip(force=True, out=out, defines=tuple(extra))
PY
