#!/bin/bash
# round 3, call d: persistent conv kernel v2 -- value check, timing, per-tile timeline
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/conv_pers_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_pers_check.txt; grep -c "bit-identical" gpurun_out/conv_pers_check.txt; grep "value check\|plan v" gpurun_out/conv_pers_check.txt
MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_tl.so timeout 600 python scripts/conv_pers_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_pers_timeline.txt; head -30 gpurun_out/conv_pers_timeline.txt
