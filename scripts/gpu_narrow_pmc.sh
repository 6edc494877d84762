#!/bin/bash
# PMC passes over scripts/narrow_prof.py (narrow conv: VALU / MFMA forms; conv11 + selection): where do the cycles go?
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/npmc; mkdir -p gpurun_out/npmc
export TMPDIR=/tmp
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/npmc/$tag" -o p -- python "$REPO/scripts/narrow_prof.py" > /dev/null 2> "$REPO/gpurun_out/npmc/$tag.err"
  echo "pmc $tag exit $?"
done
cd "$REPO"
python scripts/pmc_by_grid.py gpurun_out/npmc > gpurun_out/npmc/summary.txt 2>&1
cat gpurun_out/npmc/summary.txt | cut -c1-330
find gpurun_out/npmc -name "*.csv" -size +4M -delete
