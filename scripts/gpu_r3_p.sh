#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for t in pf nopf pf nopf; do
  if [ $t = nopf ]; then export MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_nopf.so; else unset MVSTER_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-coherent --no-other-configs --no-stream-inputs --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('deconv prefetch $t', d['value'], 'single', d['single_forward_ms'])"; done
unset MVSTER_LIB
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv11_selection or transposed" 2>&1 | tail -3
