#!/bin/bash
# round 3, call g: persistent 1x1 kernel (variant 6) -- check + timing
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/conv_pers_check.py --one 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_1x1_check.txt; cat gpurun_out/conv_1x1_check.txt
