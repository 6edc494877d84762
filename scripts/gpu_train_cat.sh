#!/bin/bash
# rocprofv3 kernel stats of training steps (config 4, one rank), steady state = long run minus short run; writes
# gpurun_out/$1/train_categories.txt and the per-kernel table train_kernels.txt
set -u
OUT=${1:-prof_train}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
REPO=$PWD
D=$REPO/gpurun_out/$OUT
rm -rf "$D/raw"; mkdir -p "$D/raw"
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/raw" -o train -- python "$REPO/scripts/train_steps.py" 512 640 5 2 6 --graph > "$D/train.json" 2> "$D/rocprof.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/raw" -o train_short -- python "$REPO/scripts/train_steps.py" 512 640 5 2 1 --graph > /dev/null 2>> "$D/rocprof.err"
cd "$REPO"
L=$(find "$D/raw" -name "train_kernel_stats.csv" | head -1); S=$(find "$D/raw" -name "train_short_kernel_stats.csv" | head -1)
cp "$L" "$D/train_kernel_stats.csv"; cp "$S" "$D/train_kernel_stats_short_run.csv"
python scripts/train_categories.py "$D/train_kernel_stats.csv" 6 "$D/train_kernel_stats_short_run.csv" 1 > "$D/train_categories.txt"
python - "$D" <<'PY' > "$D/train_kernels.txt"
import csv, sys
d = sys.argv[1]
def load(p):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(p))}
a, b = load(d + '/train_kernel_stats.csv'), load(d + '/train_kernel_stats_short_run.csv')
rows = []
for k, (c, t) in a.items():
    c0, t0 = b.get(k, (0, 0))
    if c - c0 > 0:
        rows.append(((t - t0) / 5 / 1e3, (c - c0) / 5, k))
rows.sort(reverse=True)
for t, c, k in rows:
    print("%8.1f us %6.1f x  %s" % (t, c, k[:160]))
PY
rm -rf "$D/raw"
cat "$D/train_categories.txt"
