#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
timeout 600 python scripts/conv_narrow_check.py > gpurun_out/r4_e_conv_narrow_check.txt 2>&1; tail -14 gpurun_out/r4_e_conv_narrow_check.txt
