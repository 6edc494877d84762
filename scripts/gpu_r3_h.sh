#!/bin/bash
# round 3, call h: re-tune the per-layer table with the persistent kernels as candidates; A/B the forward on one box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python scripts/conv_microbench.py --emit gpurun_out/tuning_new.json 2>&1 | grep -v amdgpu.ids > gpurun_out/tuning_log.txt; tail -3 gpurun_out/tuning_log.txt
for t in old new old new; do
  if [ $t = new ]; then export MVSTER_TUNING=$PWD/gpurun_out/tuning_new.json; else unset MVSTER_TUNING; fi
  timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$t', d['value'], 'single', d['single_forward_ms'], r['kernel'], r['frac'], r['avg_launch_us'])"
done
export MVSTER_TUNING=$PWD/gpurun_out/tuning_new.json
python scripts/layer_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/layer_table_new.txt; tail -1 gpurun_out/layer_table_new.txt
