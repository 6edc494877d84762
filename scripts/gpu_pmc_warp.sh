#!/bin/bash
# counter passes on the fused warp kernels alone (stage 3/4 shapes, wave-local vs pixel-major)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/pmc_warp; mkdir -p gpurun_out/pmc_warp
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA_[A-Z0-9_]+|TCP_[A-Z0-9_]+|SQ_[A-Z0-9_]+|TD_[A-Z0-9_]+)\b" | sort -u | tr '\n' ' ' > "$REPO/gpurun_out/pmc_warp/counters_available.txt"
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
            "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM" \
            "TA_TA_BUSY_sum TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/pmc_warp/p$i" -o p -- python "$REPO/scripts/warp_microbench.py" --eager --reps 5 --variants 3,4 --stages 3,4 > /dev/null 2> "$REPO/gpurun_out/pmc_warp/p$i.err"
  echo "pmc pass $i ($pass) exit $?"
done
cd "$REPO"
python scripts/pmc_summary.py gpurun_out/pmc_warp gpurun_out/pmc_warp/per_kernel.json > gpurun_out/pmc_warp/summary.txt 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_warp/per_kernel.json'))
for k,v in sorted(d.items()):
    if 'warp' in k:
        print(k); print('   ', {kk: round(vv,1) for kk,vv in sorted(v.items())})
PY
find gpurun_out/pmc_warp -name "*.csv" -size +4M -delete
