#!/bin/bash
# round 3, call k: fused training selection op (tests, step time), aten glue profile of the training step, bench line with top-N rooflines
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -q -x -k "train" 2>&1 | tail -4
timeout 600 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > gpurun_out/train_bench.json; python -c "
import json; d=json.load(open('gpurun_out/train_bench.json')); print('train', d['value'], d['ms_per_step'])"
timeout 600 python scripts/train_glue_profile.py --by-count 2>&1 | grep -v amdgpu.ids > gpurun_out/train_glue_profile.txt; head -60 gpurun_out/train_glue_profile.txt
timeout 400 python bench.py --no-cpu-baseline --steps 200 2>gpurun_out/bench_k.err > gpurun_out/bench_k.json; tail -2 gpurun_out/bench_k.err; python -c "
import json
d=json.load(open('gpurun_out/bench_k.json'))
print(d['value'], d['single_forward_ms'])
for r in d['rooflines']: print('  ', r['kernel'][:52], r['bound'], r['frac'], r['avg_launch_us'], r.get('share_of_timed_kernel_time'))
for r in d.get('rooflines_warp_smooth_depth', []): print('  smooth', r['kernel'][:40], r['frac'], r['avg_launch_us'])"
