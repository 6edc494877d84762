#!/bin/bash
# rocprof kernel split of the sorted warp backward at the given stages: gpu_probe_stage.sh "4 1"
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
for s in $1; do
  rm -rf /tmp/raw
  PROBE_ONLY=${2:-sorted} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/raw -o st -- python $R/scripts/warp_bwd_sorted_probe.py $s 5 2>/dev/null | tail -1
  python $R/scripts/kstats.py $(find /tmp/raw -name st_kernel_stats.csv | head -1) warp_bwd warp_agg_bwd absmax
done
