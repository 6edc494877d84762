#!/bin/bash
# round 3, call f: persistent conv kernel v4 (single-block loop body, optional static priority) -- check + timing
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/conv_pers_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_pers_check.txt; grep -c "bit-identical" gpurun_out/conv_pers_check.txt; grep "value check\|DIFFERENT\|plan v" gpurun_out/conv_pers_check.txt | head -12
