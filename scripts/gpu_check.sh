#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench (eager + hipGraph), rocprofv3 kernel stats.
# Everything is logged under gpurun_out/ (merged back into the build container).
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
nproc >> gpurun_out/device.txt; lscpu | grep -m1 "Model name" >> gpurun_out/device.txt
for f in tests/test_gpu_kernels.py tests/test_gpu_model.py; do
  timeout 600 python -m pytest $f -m gpu -q -x --timeout=300 2>&1 | tail -60 > gpurun_out/$(basename $f .py).log
  echo "== $f exit ${PIPESTATUS[0]}" | tee -a gpurun_out/summary.txt
  tail -5 gpurun_out/$(basename $f .py).log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke exit $?" | tee -a gpurun_out/summary.txt
timeout 600 python bench.py --steps 30 --warmup 5 --no-graph --no-cpu-baseline --kernel-table > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; echo "== bench eager exit $?" | tee -a gpurun_out/summary.txt
timeout 600 python bench.py --steps 50 --warmup 10 --kernel-table > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; echo "== bench graph exit $?" | tee -a gpurun_out/summary.txt
cat gpurun_out/bench_eager.json gpurun_out/bench_graph.json
tail -25 gpurun_out/bench_graph.err
