#!/bin/bash
# training-path parity tests + captured training step bench + its kernel categories
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -m gpu -x -q > gpurun_out/v_tests.txt 2>&1; tail -3 gpurun_out/v_tests.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "warp or bn or norm" > gpurun_out/v_tests_k.txt 2>&1; tail -3 gpurun_out/v_tests_k.txt
timeout 300 python bench.py --mode train --steps 20 --warmup 3 2> gpurun_out/train_v.err | cut -c1-330 | tee gpurun_out/train_v.json
bash scripts/gpu_r3_u.sh 2>&1 | head -22
