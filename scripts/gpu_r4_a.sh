#!/bin/bash
# round 4, first GPU call: baseline at the driver's arguments, the leftover questions of round 3 (captured training step at HEAD;
# do pinned H2D copies run as blit kernels?), and the full GPU test suite at HEAD
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r4_a_bench.err > gpurun_out/r4_a_bench.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_a_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','single_forward_ms','value_with_h2d')})
PY
timeout 300 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > gpurun_out/r4_train.json; cut -c1-260 gpurun_out/r4_train.json
rm -rf gpurun_out/r4_h2d; mkdir -p gpurun_out/r4_h2d
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$REPO/gpurun_out/r4_h2d" -o h2d -- \
    python "$REPO/bench.py" --steps 60 --warmup 10 --no-other-configs > "$REPO/gpurun_out/r4_h2d/line.json" 2> "$REPO/gpurun_out/r4_h2d/err.txt"
cd "$REPO"
find gpurun_out/r4_h2d -name "*_stats.csv" | while read f; do echo "== $f"; head -12 "$f" | cut -c1-200; done
find gpurun_out/r4_h2d -name "*_trace.csv" -size +8M -delete
python scripts/probes/h2d_rate.py 2>&1 | tail -12
