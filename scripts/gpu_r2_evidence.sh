#!/bin/bash
# round-2 evidence run (final kernels): PMC passes (HBM traffic with the kernel-source hash), default bench (+ CPU
# baseline), rocprofv3 stats of the bench, the other BASELINE configs, per-layer tables, training-step timings, profiles
# and roofline table, warp-kernel counters and microbenchmarks, the two atomic probes
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
bash scripts/gpu_pmc.sh > gpurun_out/pmc_run.txt 2>&1; tail -3 gpurun_out/pmc_run.txt
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py --kernel-table > gpurun_out/bench.json 2> gpurun_out/bench_kernel_table.txt; echo "bench exit $?"; cat gpurun_out/bench.json
bash scripts/gpu_profile.sh > gpurun_out/profile_run.txt 2>&1; tail -3 gpurun_out/profile_run.txt
timeout 900 python scripts/gpu_configs.py > gpurun_out/configs.jsonl 2>/dev/null; cat gpurun_out/configs.jsonl
python scripts/layer_table.py > gpurun_out/layer_table.txt 2>&1; tail -2 gpurun_out/layer_table.txt
python scripts/fpn_tail_bench.py > gpurun_out/fpn_tail_bench.txt 2>&1; tail -4 gpurun_out/fpn_tail_bench.txt
: > gpurun_out/train_steps.jsonl
for o in "" "--graph" "--coherent" "--coherent --graph"; do python scripts/train_steps.py 512 640 5 2 10 $o 2>/dev/null >> gpurun_out/train_steps.jsonl; done; cat gpurun_out/train_steps.jsonl
MVSTER_UNFUSED_ADAM=1 python scripts/train_steps.py 512 640 5 2 10 --graph 2>/dev/null >> gpurun_out/train_steps.jsonl
TRAIN_STEPS=6 bash scripts/gpu_train_profile.sh > /dev/null 2>&1; cp gpurun_out/prof_train/train_kernel_stats.csv gpurun_out/train_incoherent_kernel_stats.csv
python scripts/train_categories.py gpurun_out/train_incoherent_kernel_stats.csv 8 > gpurun_out/train_categories_incoherent.txt; cat gpurun_out/train_categories_incoherent.txt | head -14
TRAIN_STEPS="6 --coherent" bash scripts/gpu_train_profile.sh > /dev/null 2>&1; cp gpurun_out/prof_train/train_kernel_stats.csv gpurun_out/train_coherent_kernel_stats.csv
python scripts/train_categories.py gpurun_out/train_coherent_kernel_stats.csv 8 > gpurun_out/train_categories_coherent.txt; head -14 gpurun_out/train_categories_coherent.txt
python scripts/train_op_table.py > gpurun_out/train_op_table.txt 2>&1; tail -6 gpurun_out/train_op_table.txt
python scripts/train_op_table.py --coherent > gpurun_out/train_op_table_coherent.txt 2>&1; tail -5 gpurun_out/train_op_table_coherent.txt
bash scripts/gpu_pmc_warp.sh > gpurun_out/pmc_warp_run.txt 2>&1
python scripts/warp_microbench.py 2>&1 | grep -v amdgpu > gpurun_out/warp_microbench.txt; cat gpurun_out/warp_microbench.txt
python scripts/warp_bwd_probe_noise.py 2>&1 | grep -v amdgpu > gpurun_out/warp_bwd_noise.txt; cat gpurun_out/warp_bwd_noise.txt
./scripts/probes/lds_atomic_probe > gpurun_out/lds_atomic_probe.txt 2>&1
./scripts/probes/global_atomic_probe > gpurun_out/global_atomic_probe.txt 2>&1; cat gpurun_out/global_atomic_probe.txt
