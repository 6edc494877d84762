#!/bin/bash
# first GPU call of the next round: what round 3 left unmeasured (no GPU minutes left when these questions came up)
#  1. the captured training step at HEAD (the two-step plane reduction of the loss terms went in after the last evidence run)
#  2. value_with_h2d: do the pinned H2D copies run as blit kernels next to the persistent kernels?  kernel trace of the
#     default bench (a copy kernel in the list during the h2d loop answers it) -- no PMC in this pass
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > gpurun_out/r4_train.json; cut -c1-260 gpurun_out/r4_train.json
timeout 300 python bench.py --mode train --steps 20 --warmup 3 --coherent 2>/dev/null > gpurun_out/r4_train_smooth.json; cut -c1-260 gpurun_out/r4_train_smooth.json
rm -rf gpurun_out/r4_h2d; mkdir -p gpurun_out/r4_h2d
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$REPO/gpurun_out/r4_h2d" -o h2d -- \
    python "$REPO/bench.py" --steps 60 --warmup 10 --no-other-configs > "$REPO/gpurun_out/r4_h2d/line.json" 2> "$REPO/gpurun_out/r4_h2d/err.txt"
cd "$REPO"
find gpurun_out/r4_h2d -name "*_stats.csv" | while read f; do echo "== $f"; head -12 "$f" | cut -c1-200; done
find gpurun_out/r4_h2d -name "*_trace.csv" -size +8M -delete
