#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 2>&1 | tail -30 > gpurun_out/test_gpu_kernels.log
echo "== kernels exit ${PIPESTATUS[0]}"; tail -8 gpurun_out/test_gpu_kernels.log
timeout 600 python scripts/conv_microbench.py > gpurun_out/conv_microbench.txt 2>&1; echo "== microbench exit $?"
cat gpurun_out/conv_microbench.txt | cut -c1-400
timeout 600 python bench.py --steps 30 --warmup 5 --kernel-table > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "== bench exit $?"
cat gpurun_out/bench_graph.json; tail -32 gpurun_out/bench_graph.err
