#!/bin/bash
# the reworked bench line at the driver's arguments (twice: how stable is it?), and the training line
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for i in 1 2; do
  s=$(date +%s)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/r4_h_bench$i.err > gpurun_out/r4_h_bench$i.json
  echo "bench $i exit $? ($(( $(date +%s) - s )) s)"; tail -3 gpurun_out/r4_h_bench$i.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_h_bench$i.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','single_forward_ms','value_one_in_flight')})
print('timing', d['timing'])
print('h2d', d.get('value_with_h2d'))
print('train', {k:v for k,v in (d.get('train') or {}).items() if k not in ('config','rooflines')})
print('other', [(o['workload'][:30], o.get('ms_per_depth_map'), o.get('published')) for o in d.get('other_configs',[])])
print('roofline', {k:v for k,v in d['roofline'].items() if k!='per_shape'})
for r in d['rooflines'][:10]: print('   ', r['kernel'], r['bound'], r['frac'], r['avg_launch_us'], r['launches_per_step'])
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('vs_cpu_baseline'))
PY
done
timeout 600 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > gpurun_out/r4_h_train.json; cut -c1-400 gpurun_out/r4_h_train.json
