#!/bin/bash
# in-forward durations (rocprofv3 kernel trace of the eager forward) with the tree's library and the round-3 library
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
export TMPDIR=/tmp
for lib in new r03; do
  rm -rf gpurun_out/r4_d_$lib; mkdir -p gpurun_out/r4_d_$lib
  if [ $lib = r03 ]; then export MVSTER_LIB=$REPO/mvster_amd/csrc/ab/libmvster_r03.so MVSTER_LIB_LAX=1; fi
  cd /tmp
  MVSTER_NO_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/r4_d_$lib" -o t -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-graph --no-cpu-baseline --no-coherent --no-other-configs --no-stream-inputs > /dev/null 2> "$REPO/gpurun_out/r4_d_$lib/err.txt"
  cd "$REPO"
  echo "== $lib"; grep -E "conv_small|conv_narrow|deconv_select|fpn_|warp_agg" gpurun_out/r4_d_$lib/t_kernel_stats.csv | cut -d, -f1-7 | sed 's/(anonymous namespace):://g; s/void //' | cut -c1-150
  find gpurun_out/r4_d_$lib -name "*_trace.csv" -size +6M -delete
done
