#!/bin/bash
# round 3, call o: FPN gather with batched staging loads: tests, isolated timings, bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fpn" 2>&1 | tail -4
timeout 300 python scripts/fpn_tail_bench.py 2>&1 | grep -v amdgpu.ids | grep "gather\|whole\|lateral" | tee gpurun_out/fpn_tail_bench.txt
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-coherent --no-other-configs --no-stream-inputs --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], 'single', d['single_forward_ms'])"; done
