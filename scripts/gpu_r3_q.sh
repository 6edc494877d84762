#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fpn" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "plans_follow or golden_teacher" 2>&1 | tail -2
timeout 300 python scripts/fpn_tail_bench.py 2>&1 | grep "lateral\|whole"
for fl in "" "--inflight 3" "" "--inflight 3"; do timeout 300 python bench.py --no-cpu-baseline --no-coherent --no-other-configs --no-stream-inputs --steps 200 $fl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench $fl', d['value'], 'single', d['single_forward_ms'])"; done
