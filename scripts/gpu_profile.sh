#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (no PMC here; counters get their own pass).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof" -o bench -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-graph --no-cpu-baseline --no-coherent --no-overlap --no-train --no-other-configs --no-stream-inputs > "$REPO/gpurun_out/prof/bench_under_rocprof.json" 2> "$REPO/gpurun_out/prof/rocprof.err"
echo "rocprof exit $?"
cd "$REPO"
ls gpurun_out/prof | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" | cut -c1-200
# keep the merge small: the per-dispatch trace can be tens of MB
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
