#!/bin/bash
# round 3, call s: Winograd kernels -- tests, re-tune every layer with the new candidates, bench with the new table
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "winograd or pingpong or persistent or untuned or upsample_add" 2>&1 | tail -3
timeout 1800 python scripts/conv_microbench.py --emit gpurun_out/tuning_new.json 2>&1 | grep -v amdgpu.ids > gpurun_out/tuning_log.txt; tail -3 gpurun_out/tuning_log.txt
MVSTER_TUNING=$PWD/gpurun_out/tuning_new.json timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-coherent > gpurun_out/bench_wino.json 2> gpurun_out/bench_wino.err; python -c "
import json; d=json.load(open('gpurun_out/bench_wino.json')); print(d['value'], d['ms_per_step'], d.get('single_forward_ms'))"
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-coherent > gpurun_out/bench_old.json 2> gpurun_out/bench_old.err; python -c "
import json; d=json.load(open('gpurun_out/bench_old.json')); print(d['value'], d['ms_per_step'], d.get('single_forward_ms'))"
