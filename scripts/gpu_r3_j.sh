#!/bin/bash
# round 3, call j: fused FPN conv0, graphed-step gradient test, bucketed all-reduce test, bench A/B (fused conv0 on/off), train bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "persistent" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -k "graphed_step or sinkhorn" 2>&1 | tail -8
timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null > gpurun_out/bench_j.json; python -c "
import json
d=json.load(open('gpurun_out/bench_j.json')); r=d['roofline']
print(d['value'], 'single', d['single_forward_ms'], r['kernel'], r['frac'], r['avg_launch_us'], d.get('value_with_h2d'), d.get('other_configs'))"
timeout 600 python bench.py --mode train --steps 20 --warmup 3 2>gpurun_out/train_bench.err > gpurun_out/train_bench.json; tail -3 gpurun_out/train_bench.err; cut -c1-1500 gpurun_out/train_bench.json
timeout 600 python bench.py --mode train --steps 20 --warmup 3 --coherent 2>/dev/null > gpurun_out/train_bench_coherent.json; cut -c1-400 gpurun_out/train_bench_coherent.json
MVSTER_FORCE_BUCKET=1 timeout 600 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > gpurun_out/train_bench_bucket.json; cut -c1-400 gpurun_out/train_bench_bucket.json
