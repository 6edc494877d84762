#!/bin/bash
# round 3, call i: full GPU test suite with the re-tuned table (persistent kernels in the plans), layer table, bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_tests.sh
python scripts/layer_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/layer_table.txt; tail -1 gpurun_out/layer_table.txt
timeout 300 python bench.py --no-cpu-baseline --kernel-table 2> gpurun_out/bench_kernel_table.txt > gpurun_out/bench_i.json; cut -c1-300 gpurun_out/bench_i.json; head -12 gpurun_out/bench_kernel_table.txt
