#!/bin/bash
# out-of-window taps through the wave-local queue: parity of the backward, time vs depth noise, training steps
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "backward" 2>&1 | tail -6
python scripts/warp_bwd_probe_noise.py 2>&1 | tail -24
for mode in "--graph"; do
  python scripts/train_steps.py 512 640 5 2 8 $mode 2>&1 | grep config
  python scripts/train_steps.py 512 640 5 2 8 $mode --coherent 2>&1 | grep config
done
python scripts/train_steps.py 512 640 5 2 8 2>&1 | grep config
