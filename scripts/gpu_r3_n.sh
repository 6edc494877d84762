#!/bin/bash
# round 3, call n: fused conv11 + selection: tests, model goldens, A/B bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv11_selection or select_depth" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -k "fused_stage_selection" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -5
for t in on off on off; do
  if [ $t = off ]; then export MVSTER_NO_FUSE_SELECT=1; else unset MVSTER_NO_FUSE_SELECT; fi
  timeout 300 python bench.py --no-cpu-baseline --no-coherent --no-other-configs --no-stream-inputs --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fused select $t', d['value'], 'single', d['single_forward_ms'])"
done
