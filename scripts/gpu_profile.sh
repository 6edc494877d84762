#!/bin/bash
# rocprofv3 kernel-trace stats of the bench command (no PMC here; counters get their own pass).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-graph --no-cpu-baseline > "$OLDPWD/gpurun_out/prof/bench_under_rocprof.json" 2> "$OLDPWD/gpurun_out/prof/rocprof.err"
echo "rocprof exit $?"
cd "$OLDPWD"
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
