#!/bin/bash
# rocprofv3 kernel stats of a few training steps (config 4, one rank)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/prof_train; mkdir -p gpurun_out/prof_train
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train" -o train -- python "$REPO/scripts/train_steps.py" 512 640 5 2 ${TRAIN_STEPS:-3} > "$REPO/gpurun_out/prof_train/train.json" 2> "$REPO/gpurun_out/prof_train/rocprof.err"
echo "rocprof exit $?"
# a shorter run of the same script: train_categories.py subtracts it (start-up work is not per-step work)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_train" -o train_short -- python "$REPO/scripts/train_steps.py" 512 640 5 2 ${TRAIN_STEPS_SHORT:-1} > /dev/null 2>> "$REPO/gpurun_out/prof_train/rocprof.err"
echo "rocprof (short run) exit $?"
cd "$REPO"
cat gpurun_out/prof_train/train.json
f=gpurun_out/prof_train/train_kernel_stats.csv
[ -n "$f" ] && head -30 "$f" | cut -c1-220
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
