#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=300 -k "warp" 2>&1 | tail -4
timeout 200 python scripts/warp_microbench.py --variants 3,4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/warp_microbench.txt
for i in 1 2; do
  timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_new_$i.json 2>/dev/null
  MVSTER_LIB=$PWD/mvster_amd/csrc/ab/libmvster_r01.so MVSTER_LIB_LAX=1 timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/bench_r01lib_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_new_[12].json')+glob.glob('gpurun_out/bench_r01lib_[12].json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        w=[r for r in d['rooflines'] if 'warp' in r['kernel']]
        print(f, d['value'], d['single_forward_ms'], d['roofline']['avg_launch_us'], [(r['kernel'][-22:], r['avg_launch_us']) for r in w])
    except Exception as e: print(f, 'ERR', e)
PY
