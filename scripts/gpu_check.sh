#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for f in tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_fusion.py; do
  timeout 600 python -m pytest $f -m gpu -q --timeout=300 2>&1 | tail -30 > gpurun_out/$(basename $f .py).log
  echo "== $f exit ${PIPESTATUS[0]}"; tail -4 gpurun_out/$(basename $f .py).log
done
timeout 600 python bench.py --steps 100 --warmup 10 --kernel-table > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "== bench exit $?"
cat gpurun_out/bench_graph.json; tail -36 gpurun_out/bench_graph.err
