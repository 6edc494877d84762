#!/bin/bash
# training-path parity tests + captured training step bench + its kernel categories
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --mode train --steps 20 --warmup 3 2> gpurun_out/train_v.err | tee gpurun_out/train_v.json
bash scripts/gpu_r3_u.sh
