#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/pmc_bwd; mkdir -p gpurun_out/pmc_bwd
export TMPDIR=/tmp
cd /tmp
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INSTS_GDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  MVSTER_BWD_ATOMIC=1 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/pmc_bwd/p$i" -o p -- python "$REPO/scripts/warp_bwd_probe.py" > /dev/null 2> "$REPO/gpurun_out/pmc_bwd/p$i.err"
  echo "pmc pass $i exit $?"
done
cd "$REPO"
python scripts/pmc_summary.py gpurun_out/pmc_bwd gpurun_out/pmc_bwd/per_kernel.json > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_bwd/per_kernel.json'))
for k,v in sorted(d.items()):
    if 'bwd' in k:
        print(k); print('   ', {kk: round(vv,1) for kk,vv in sorted(v.items())})
PY
find gpurun_out/pmc_bwd -name "*.csv" -size +2M -delete
