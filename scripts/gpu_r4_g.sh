#!/bin/bash
# tests at HEAD (kernels + model), then A/B: output stores of the persistent conv kernels written through (sc1) vs plain
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for f in tests/test_gpu_kernels.py tests/test_gpu_model.py; do
  timeout 1200 python -m pytest $f -m gpu -q --timeout=900 2>&1 | tail -8 > gpurun_out/r4_g_$(basename $f .py).log
  echo "== $f"; tail -3 gpurun_out/r4_g_$(basename $f .py).log
done
for rep in 1 2; do
for lib in plain wt; do
  if [ $lib = wt ]; then export MVSTER_LIB=$PWD/mvster_amd/csrc/ab/libmvster_wt.so; else unset MVSTER_LIB; fi
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-stream-inputs --no-coherent 2>gpurun_out/r4_g_bench_$lib.err > gpurun_out/r4_g_bench_$lib.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_g_bench_$lib.json').read().strip().splitlines()[-1])
print('$lib', {k:d.get(k) for k in ('value','ms_per_step','single_forward_ms')})
PY
done
done
