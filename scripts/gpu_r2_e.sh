#!/bin/bash
# re-associated FPN mid level: parity of the touched paths, bench, kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -k "fpn or conv" 2>&1 | tail -8
python -m pytest tests/test_gpu_model.py -m gpu -q --timeout=900 -x 2>&1 | tail -8
python bench.py --no-cpu-baseline | tee gpurun_out/bench_e.json
bash scripts/gpu_profile.sh 2>&1 | tail -45
