#!/bin/bash
# round 3, call c: per-tile timeline of the persistent conv kernel at 1..4 workgroups per CU
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_tl.so timeout 600 python scripts/conv_pers_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_pers_timeline.txt; cat gpurun_out/conv_pers_timeline.txt
