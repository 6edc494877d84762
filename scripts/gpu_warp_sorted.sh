#!/bin/bash
# per-kernel split of the sorted warp backward at the four stages (rocprofv3 kernel stats, 5 repetitions each)
set -u
OUT=${1:-warp_sorted}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
D=$REPO/gpurun_out/$OUT
mkdir -p "$D"
export TMPDIR=/tmp
: > "$D/summary.txt"
for s in 1 2 3 4; do
  python scripts/warp_bwd_sorted_probe.py $s >> "$D/summary.txt" 2>/dev/null
  rm -rf "$D/raw"; mkdir -p "$D/raw"
  (cd /tmp && PROBE_ONLY=sorted timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/raw" -o st -- python "$REPO/scripts/warp_bwd_sorted_probe.py" $s 5 > /dev/null 2>&1)
  f=$(find "$D/raw" -name "st_kernel_stats.csv" | head -1)
  python - "$f" $s >> "$D/summary.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Name"] for k in ("warp_", "absmax"))]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "fwd" in r["Name"]:
        continue
    print("   stage %s  %8.1f us avg  %3d x  %s" % (sys.argv[2], float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"][:110]))
PY
done
rm -rf "$D/raw"
cat "$D/summary.txt"
