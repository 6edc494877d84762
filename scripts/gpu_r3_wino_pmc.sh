#!/bin/bash
# SQ counters of the Winograd kernel (variant 8) next to the persistent direct kernel (variant 5) on the 16 -> 16 layer.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
OUT=$REPO/gpurun_out/wino_pmc; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  for v in 520 517; do
    timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag.$v -o p -- python $REPO/scripts/conv_one.py 16 16 1 3 1 5 1 256 320 2 1 $v 10 > /dev/null 2> $OUT/$tag.$v.err
    echo "pmc $tag v$v exit $?"
  done
done
cd $REPO
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/wino_pmc/*/*counter_collection.csv") + glob.glob("gpurun_out/wino_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_wino" in k or "conv_pers" in k:
            rows[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, m in rows.items():
    print(k)
    for c in sorted(m):
        v = m[c]
        print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
find gpurun_out/wino_pmc -name "*.csv" -size +2M -delete
