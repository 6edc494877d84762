#!/bin/bash
# kernel stats of the CAPTURED training step (bench.py --mode train replays one hipGraph per step)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/prof_trainbench; mkdir -p gpurun_out/prof_trainbench
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_trainbench" -o tb -- python "$REPO/bench.py" --mode train --steps 10 --warmup 2 > "$REPO/gpurun_out/prof_trainbench/line.json" 2> "$REPO/gpurun_out/prof_trainbench/err.txt"
cd "$REPO"
f=$(find gpurun_out/prof_trainbench -name "*kernel_stats.csv" | head -1)
python scripts/train_categories.py $f 12 | head -40
find gpurun_out/prof_trainbench -name "*kernel_trace.csv" -delete
