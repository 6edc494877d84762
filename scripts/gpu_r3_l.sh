#!/bin/bash
# round 3, call l: LDS-window warp kernel (variant 5): bit-identity tests, microbench vs the wave kernel in both depth regimes
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "lds_window" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "fused_stage_selection" 2>&1 | tail -25
timeout 600 python scripts/warp_window_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/warp_window_bench.txt
