#!/bin/bash
# BatchNorm reductions finished in-kernel, fused Adam: parity of the training passes, then step timings
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_train.py -m gpu -q --timeout=900 -x 2>&1 | tail -8
for mode in "" "--graph"; do
  python scripts/train_steps.py 512 640 5 2 8 $mode
  python scripts/train_steps.py 512 640 5 2 8 $mode --coherent
done
MVSTER_UNFUSED_ADAM=1 python scripts/train_steps.py 512 640 5 2 8 --graph
