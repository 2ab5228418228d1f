#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 4
bash tools/r4_variants.sh r4_k2b "c3 c2 c3real c4" default
