#!/bin/bash
# write-through output stores: kernel + model tests, then the bench line with the tree's library and its write-back twin
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
for f in tests/test_gpu_kernels.py tests/test_gpu_model.py; do
  timeout 1200 python -m pytest $f -m gpu -q -x --timeout=900 2>&1 | tail -8 > gpurun_out/r4_f_$(basename $f .py).log
  echo "== $f"; tail -3 gpurun_out/r4_f_$(basename $f .py).log
done
for rep in 1 2; do
for lib in wt wb; do
  if [ $lib = wb ]; then export MVSTER_LIB=$PWD/mvster_amd/csrc/ab/libmvster_wb.so; else unset MVSTER_LIB; fi
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-stream-inputs --no-coherent 2>gpurun_out/r4_f_bench_$lib.err > gpurun_out/r4_f_bench_$lib.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_f_bench_$lib.json').read().strip().splitlines()[-1])
print('$lib', {k:d.get(k) for k in ('value','ms_per_step','single_forward_ms')})
PY
done
done
unset MVSTER_LIB
timeout 600 python scripts/conv_narrow_check.py 2>&1 | tail -12 > gpurun_out/r4_f_conv_narrow_check.txt; cat gpurun_out/r4_f_conv_narrow_check.txt
