#!/bin/bash
# round 3, call e: persistent conv kernel v3 -- value check + timing; SQ counters of the 16->16 layer at 1 / 2 / 3 workgroups per CU
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/conv_pers_check.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_pers_check.txt; grep -c "bit-identical" gpurun_out/conv_pers_check.txt; grep "value check\|DIFFERENT\|plan v" gpurun_out/conv_pers_check.txt | head -24
rm -rf gpurun_out/pmc_pers; mkdir -p gpurun_out/pmc_pers
cd /tmp
for wpc in 1 2 3; do
  var=$((5 + wpc * 256))
  i=0
  for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
              "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/pmc_pers/w${wpc}p$i" -o p -- python "$REPO/scripts/conv_one.py" 16 16 1 3 1 5 1 256 320 2 1 $var 10 > /dev/null 2> "$REPO/gpurun_out/pmc_pers/w${wpc}p$i.err"
  done
done
# the LDS-staged kernel for comparison
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
            "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/pmc_pers/ldsp$i" -o p -- python "$REPO/scripts/conv_one.py" 16 16 1 3 1 5 1 256 320 2 1 1 10 > /dev/null 2> "$REPO/gpurun_out/pmc_pers/ldsp$i.err"
done
cd "$REPO"
python - <<'PY'
import csv, glob, os
from collections import defaultdict
for tag in ("w1", "w2", "w3", "lds"):
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/pmc_pers/%sp*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_" not in r["Kernel_Name"]:
                continue
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    m = {k: v[0] / max(v[1], 1) for k, v in agg.items()}
    if not m:
        print(tag, "no data"); continue
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print("%s: waves %d  wave_cycles %.3g  busy_cycles %.3g  gui_active %.3g | of wave cycles: wait_any %.1f%% wait_inst %.1f%% active_any %.1f%% (valu %.1f%% sca %.1f%% lds %.1f%% vmem %.1f%% misc %.1f%%)  inst_cycles_vmem %.3g  wait_inst_lds %.1f%%  mfma_busy %.3g valu_insts/wave %.0f" % (
        tag, m.get("SQ_WAVES", 0), wc, m.get("SQ_BUSY_CYCLES", 0), m.get("GRBM_GUI_ACTIVE", 0),
        100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_SCA", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
        100 * m.get("SQ_ACTIVE_INST_VMEM", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_MISC", 0) / wc, m.get("SQ_INST_CYCLES_VMEM", 0),
        100 * m.get("SQ_WAIT_INST_LDS", 0) / wc, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_WAVES", 1), 1)))
PY
find gpurun_out/pmc_pers -name "*.csv" -size +2M -delete
