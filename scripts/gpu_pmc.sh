#!/bin/bash
# PMC passes (own runs, kernel-trace only): HBM traffic and SQ activity per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/pmc/$tag" -o p -- python "$REPO/bench.py" --steps 3 --warmup 2 --no-graph --no-cpu-baseline --no-coherent --no-overlap --no-train > /dev/null 2> "$REPO/gpurun_out/pmc/$tag.err"
  echo "pmc $tag exit $?"
done
cd "$REPO"
python scripts/pmc_summary.py gpurun_out/pmc gpurun_out/pmc/pmc_per_kernel.json gpurun_out/pmc/pmc_traffic.json > gpurun_out/pmc/summary.txt 2>&1
head -30 gpurun_out/pmc/summary.txt | cut -c1-250
find gpurun_out/pmc -name "*.csv" -size +8M -delete
