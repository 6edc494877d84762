#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "persistent" 2>&1 | tail -2
timeout 600 python scripts/conv_pers_check.py --time-only 2>&1 | grep "64->64 3x3 \|64->32 3x3\|32->32 3x3 "
