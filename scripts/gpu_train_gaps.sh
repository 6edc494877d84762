#!/bin/bash
# kernel trace of the captured training step -> idle time between its kernels (scripts/train_gaps.py)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
D=$REPO/gpurun_out/gaps
rm -rf "$D"; mkdir -p "$D/raw"
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$D/raw" -o train -- python "$REPO/scripts/train_steps.py" 512 640 5 2 4 --graph > "$D/train.json" 2> "$D/rocprof.err"
cd "$REPO"
T=$(find "$D/raw" -name "train_kernel_trace.csv" | head -1)
python scripts/train_gaps.py "$T" ${1:-536} | tee "$D/train_gaps.txt"
rm -rf "$D/raw"
