#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -k "backward" 2>&1 | tail -8
python -m pytest tests/test_gpu_train.py -m gpu -q --timeout=600 2>&1 | tail -5
echo "--- train step, gather backward"
python scripts/train_steps.py 512 640 5 2 8
echo "--- train step, atomic backward"
MVSTER_BWD_ATOMIC=1 python scripts/train_steps.py 512 640 5 2 8
bash scripts/gpu_train_profile.sh 2>&1 | tail -40
