#!/bin/bash
# all GPU tests, one log per file (no -x: every failure of a run is reported)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for f in tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_fusion.py; do
  s=$(date +%s)
  timeout 1200 python -m pytest $f -m gpu -q --timeout=900 --durations=6 2>&1 | tail -60 > gpurun_out/$(basename $f .py).log
  echo "== $f exit ${PIPESTATUS[0]} ($(( $(date +%s) - s )) s)"; grep -E "passed|failed|error" gpurun_out/$(basename $f .py).log | tail -3
  grep -E "^(FAILED|ERROR)" gpurun_out/$(basename $f .py).log | head -20
done
# the kernel forms kept for the record (probe library: make -C mvster_amd/csrc probes)
if [ -f mvster_amd/csrc/libmvster_hip_probes.so ]; then
  MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_probes.so timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=600 -k "variants_bit_identical or lds_window or pingpong or warp_agg_forward or bf16_split" 2>&1 | tail -5 > gpurun_out/test_gpu_probes.log
  echo "== probe library"; tail -2 gpurun_out/test_gpu_probes.log
fi
