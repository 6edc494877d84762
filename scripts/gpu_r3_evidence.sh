#!/bin/bash
# round-3 evidence run (final kernels): PMC passes (HBM traffic with the kernel-source hash), default bench (+ CPU baseline),
# rocprofv3 stats of the bench, per-layer table, training bench lines + kernel categories, full GPU test suite
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
bash scripts/gpu_pmc.sh > gpurun_out/pmc_run.txt 2>&1; tail -3 gpurun_out/pmc_run.txt
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python bench.py --kernel-table > gpurun_out/bench.json 2> gpurun_out/bench_kernel_table.txt; echo "bench exit $?"; cut -c1-300 gpurun_out/bench.json
bash scripts/gpu_profile.sh > gpurun_out/profile_run.txt 2>&1; tail -3 gpurun_out/profile_run.txt
python scripts/layer_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/layer_table.txt; tail -1 gpurun_out/layer_table.txt
timeout 600 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > gpurun_out/train_bench.json; cut -c1-200 gpurun_out/train_bench.json
timeout 600 python bench.py --mode train --steps 20 --warmup 3 --coherent 2>/dev/null > gpurun_out/train_bench_smooth.json; cut -c1-200 gpurun_out/train_bench_smooth.json
timeout 600 python bench.py --mode train --steps 20 --warmup 3 --no-graph 2>/dev/null > gpurun_out/train_bench_eager.json; cut -c1-200 gpurun_out/train_bench_eager.json
TRAIN_STEPS=6 bash scripts/gpu_train_profile.sh > /dev/null 2>&1; cp gpurun_out/prof_train/train_kernel_stats.csv gpurun_out/train_kernel_stats.csv
python scripts/train_categories.py gpurun_out/train_kernel_stats.csv 8 > gpurun_out/train_categories.txt; head -16 gpurun_out/train_categories.txt
bash scripts/gpu_tests.sh
