#!/bin/bash
# round 3, call b: persistent LDS-DMA conv kernel -- value check vs the direct kernel, timing vs the plan's choice
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/conv_pers_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_pers_check.txt; tail -60 gpurun_out/conv_pers_check.txt
