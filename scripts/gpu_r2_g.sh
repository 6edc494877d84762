#!/bin/bash
# training FPN finest level re-associated: kernel + module parity, then step timings
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -k "fpn" 2>&1 | tail -8
python -m pytest tests/test_gpu_train.py -m gpu -q --timeout=900 2>&1 | tail -12
for mode in "" "--graph"; do
  python scripts/train_steps.py 512 640 5 2 8 $mode 2>&1 | grep config
  python scripts/train_steps.py 512 640 5 2 8 $mode --coherent 2>&1 | grep config
done
