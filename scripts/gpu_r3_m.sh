#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for wv in 3 4; do echo "== tile kernel at min $wv waves/SIMD"; MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_w$wv.so timeout 300 python scripts/warp_window_bench.py 2>&1 | grep "^stage [34]"; done
