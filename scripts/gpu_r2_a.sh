#!/bin/bash
# round 2, run A: all GPU tests, default bench, warp microbench (pixel-major kernel at 1/4 waves per workgroup),
# A/B of the bench against the round-1 library on the same box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/device.txt
for f in tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_fusion.py; do
  timeout 900 python -m pytest $f -m gpu -q --timeout=600 --durations=8 2>&1 | tail -40 > gpurun_out/$(basename $f .py).log
  echo "== $f exit ${PIPESTATUS[0]}"; tail -14 gpurun_out/$(basename $f .py).log
done
for nw in 1 4; do
  MVSTER_PIX_NW=$nw timeout 300 python scripts/warp_microbench.py > gpurun_out/warp_microbench_nw$nw.txt 2>&1; cat gpurun_out/warp_microbench_nw$nw.txt
done
timeout 600 python bench.py --steps 200 --warmup 20 --kernel-table > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "== bench exit $?"
cat gpurun_out/bench.json; tail -40 gpurun_out/bench.err
MVSTER_LIB=$PWD/mvster_amd/csrc/ab/libmvster_r01.so MVSTER_LIB_LAX=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --kernel-table > gpurun_out/bench_r01lib.json 2> gpurun_out/bench_r01lib.err; echo "== bench(r01 lib) exit $?"
cat gpurun_out/bench_r01lib.json; tail -40 gpurun_out/bench_r01lib.err
