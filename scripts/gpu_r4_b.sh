#!/bin/bash
# round 4, second GPU call: the two new MFMA kernels (narrow 3x3 conv, conv11 + selection) -- parity tests first, then
# isolated timings against the VALU kernels (narrow: variant 3 in the same library; deconv_select: the round-3 library), then
# the bench line with both libraries on this box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "narrow_mfma or fused_conv11 or conv_bn_relu" > gpurun_out/r4_b_tests.txt 2>&1
echo "tests exit $?"; tail -15 gpurun_out/r4_b_tests.txt
timeout 600 python scripts/conv_narrow_check.py > gpurun_out/r4_b_conv_narrow_check.txt 2>&1; tail -16 gpurun_out/r4_b_conv_narrow_check.txt
timeout 300 python scripts/deconv_select_check.py > gpurun_out/r4_b_deconv_select_new.txt 2>&1; tail -9 gpurun_out/r4_b_deconv_select_new.txt
MVSTER_LIB=$PWD/mvster_amd/csrc/ab/libmvster_r03.so MVSTER_LIB_LAX=1 timeout 300 python scripts/deconv_select_check.py > gpurun_out/r4_b_deconv_select_r03.txt 2>&1; tail -9 gpurun_out/r4_b_deconv_select_r03.txt
for lib in new r03; do
  if [ $lib = r03 ]; then export MVSTER_LIB=$PWD/mvster_amd/csrc/ab/libmvster_r03.so MVSTER_LIB_LAX=1; fi
  timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-stream-inputs 2>gpurun_out/r4_b_bench_$lib.err > gpurun_out/r4_b_bench_$lib.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_b_bench_$lib.json').read().strip().splitlines()[-1])
print('$lib', {k:d.get(k) for k in ('value','ms_per_step','single_forward_ms')}, d['roofline']['kernel'], d['roofline']['frac'])
PY
done
