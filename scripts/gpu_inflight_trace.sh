#!/bin/bash
# kernel trace of the two-in-flight steady state + its overlap analysis (scripts/inflight_steady_state.py)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/inflight; mkdir -p gpurun_out/inflight
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/inflight" -o t -- python "$REPO/scripts/inflight_steady_state.py" run > /dev/null 2> "$REPO/gpurun_out/inflight/rocprof.err"
echo "rocprof exit $?"
cd "$REPO"
f=$(find gpurun_out/inflight -name "*kernel_trace.csv" | head -1)
python scripts/inflight_steady_state.py --analyse "$f" | tee gpurun_out/inflight/steady_state.txt
gzip -9 "$f"
