#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
TRAIN_STEPS=6 bash scripts/gpu_train_profile.sh > gpurun_out/train_profile.log 2>&1
f=$(find gpurun_out/prof_train -name "*kernel_stats.csv" | head -1)
python scripts/train_categories.py $f 8
