#!/bin/bash
# launch-shape sweep of the pixel-major warp kernel (hypotheses per lane x occupancy target) at the two fine stages
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/warp_sweep.txt
for dpl in 2 0; do for wpe in 4 5 6; do
  echo "== MVSTER_PIX_DPL=$dpl MVSTER_PIX_WPE=$wpe" >> gpurun_out/warp_sweep.txt
  MVSTER_PIX_DPL=$dpl MVSTER_PIX_WPE=$wpe timeout 200 python scripts/warp_microbench.py --variants 3,4 --stages 2,3,4 2>&1 | grep -v amdgpu.ids >> gpurun_out/warp_sweep.txt
done; done
cat gpurun_out/warp_sweep.txt
