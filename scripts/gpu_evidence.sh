#!/bin/bash
# Evidence run at the current kernel hash (tag = $1, e.g. r04_d): PMC passes (HBM traffic per kernel, gated on the source hash),
# the bench line at the driver's arguments and at 200-step windows, rocprofv3 kernel stats, the per-layer table, the training
# bench lines + kernel categories, the full GPU test suite.  Summaries are copied to profiles/<tag>_*.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=${1:-r04_x}
mkdir -p gpurun_out/ev
export TMPDIR=/tmp
E=gpurun_out/ev
rocm-smi --showproductname 2>/dev/null | head -8 > $E/${TAG}_device.txt
bash scripts/gpu_pmc.sh > $E/pmc_run.txt 2>&1; tail -3 $E/pmc_run.txt
cp gpurun_out/pmc/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/pmc/pmc_traffic.json $E/pmc_traffic.json
cp gpurun_out/pmc/summary.txt $E/${TAG}_pmc_summary.txt
cp gpurun_out/pmc/pmc_per_kernel.json $E/${TAG}_pmc_per_kernel.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --kernel-table > $E/${TAG}_bench_driver_args.json 2> $E/${TAG}_kernel_table_hip_events.txt; echo "bench (driver arguments) exit $?"; cut -c1-260 $E/${TAG}_bench_driver_args.json
timeout 900 python bench.py --no-cpu-baseline --no-train --no-other-configs > $E/${TAG}_bench_200steps.json 2>/dev/null; cut -c1-200 $E/${TAG}_bench_200steps.json
bash scripts/gpu_profile.sh > $E/profile_run.txt 2>&1; tail -3 $E/profile_run.txt
cp $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) $E/${TAG}_rocprof_kernel_stats.csv
cp gpurun_out/prof/bench_under_rocprof.json $E/${TAG}_bench_under_rocprof.json
python scripts/layer_table.py 2>&1 | grep -v amdgpu.ids > $E/${TAG}_layer_table.txt; tail -1 $E/${TAG}_layer_table.txt
timeout 600 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > $E/${TAG}_train_bench.json; cut -c1-200 $E/${TAG}_train_bench.json
timeout 600 python bench.py --mode train --steps 20 --warmup 3 --coherent 2>/dev/null > $E/${TAG}_train_bench_smooth_depth.json; cut -c1-200 $E/${TAG}_train_bench_smooth_depth.json
# training categories of the CAPTURED step (what the bench replays): 6 replays minus 1, kernel by kernel
timeout 900 bash scripts/gpu_train_cat.sh evtrain > /dev/null 2>&1
cp gpurun_out/evtrain/train_kernel_stats.csv $E/${TAG}_train_kernel_stats.csv
cp gpurun_out/evtrain/train_kernel_stats_short_run.csv $E/${TAG}_train_kernel_stats_short_run.csv
cp gpurun_out/evtrain/train_categories.txt $E/${TAG}_train_categories.txt; head -18 $E/${TAG}_train_categories.txt
cp gpurun_out/evtrain/train_kernels.txt $E/${TAG}_train_kernels.txt
timeout 600 bash scripts/gpu_warp_sorted.sh evwarp > /dev/null 2>&1; cp gpurun_out/evwarp/summary.txt $E/${TAG}_warp_bwd_sorted_vs_atomic.txt
timeout 600 bash scripts/gpu_inflight_trace.sh > /dev/null 2>&1; cp gpurun_out/inflight/steady_state.txt $E/${TAG}_inflight_steady_state.txt
bash scripts/gpu_tests.sh
for f in kernels model train fusion; do cp gpurun_out/parity_$f.json $E/${TAG}_parity_$f.json 2>/dev/null; done
cp gpurun_out/test_gpu_*.log $E/ 2>/dev/null
