#!/bin/bash
# PMC passes (own runs, kernel-trace only): SQ activity and HBM traffic per kernel.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)?\s*:?\s*(SQ_|TCC_|GRBM_|FETCH|WRITE|MfmaUtil|VALUBusy)" | head -0
rocprofv3 -L > "$REPO/gpurun_out/pmc/counters.txt" 2>&1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/pmc/$tag" -o p -- python "$REPO/bench.py" --steps 3 --warmup 2 --no-graph --no-cpu-baseline > /dev/null 2> "$REPO/gpurun_out/pmc/$tag.err"
  echo "pmc $tag exit $?"
done
cd "$REPO"
python scripts/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt 2>&1
head -60 gpurun_out/pmc/summary.txt
find gpurun_out/pmc -name "*.csv" -size +8M -delete
