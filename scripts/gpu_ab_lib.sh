#!/bin/bash
# A/B of two builds of the library on one box: gpu_ab_lib.sh <other .so under mvster_amd/csrc/ab/> [extra per-layer script]
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OTHER=$PWD/mvster_amd/csrc/ab/$1
if [ $# -gt 1 ]; then
  echo "== tree library"; (cd scripts && timeout 600 python $2 --time-only 2>&1 | grep -v amdgpu.ids | tail -${3:-30})
  echo "== $1"; (cd scripts && MVSTER_LIB=$OTHER MVSTER_LIB_LAX=1 timeout 600 python $2 --time-only 2>&1 | grep -v amdgpu.ids | tail -${3:-30})
fi
for rep in 1 2 3; do
for lib in tree other; do
  if [ $lib = other ]; then export MVSTER_LIB=$OTHER MVSTER_LIB_LAX=1; else unset MVSTER_LIB MVSTER_LIB_LAX; fi
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-stream-inputs --no-coherent --no-train 2>/dev/null > gpurun_out/ab_$lib.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/ab_$lib.json').read().strip().splitlines()[-1])
print('$lib', {k:d.get(k) for k in ('value','ms_per_step','single_forward_ms')})
PY
done
done
