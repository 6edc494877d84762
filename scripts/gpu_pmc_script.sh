#!/bin/bash
# PMC passes over any script: gpu_pmc_script.sh "<python args>" <kernel name prefixes, comma separated>
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
rm -rf gpurun_out/spmc; mkdir -p gpurun_out/spmc
export TMPDIR=/tmp
cd /tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd "$REPO" && timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$REPO/gpurun_out/spmc/$tag" -o p -- python $1 > /dev/null 2> "$REPO/gpurun_out/spmc/$tag.err")
  echo "pmc $tag exit $?"
done
cd "$REPO"
python scripts/pmc_by_grid.py gpurun_out/spmc "$2" > gpurun_out/spmc/summary.txt 2>&1
cat gpurun_out/spmc/summary.txt | cut -c1-330
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/spmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'fpn' in k or 'narrow' in k:
            agg[k.split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value'] or 0))
for k, m in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in m.items() if c.startswith('SQ_INSTS') or c == 'SQ_WAIT_INST_LDS' or c == 'SQ_WAVES'})
PY
find gpurun_out/spmc -name "*.csv" -size +4M -delete
