#!/bin/bash
# Python-side changes only: training parity tests, training bench lines, default bench line (value_with_h2d)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -x -q > gpurun_out/w_tests_train.txt 2>&1; tail -2 gpurun_out/w_tests_train.txt
timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "train_step or data_parallel" > gpurun_out/w_tests_model.txt 2>&1; tail -2 gpurun_out/w_tests_model.txt
timeout 300 python bench.py --mode train --steps 20 --warmup 3 2>/dev/null > gpurun_out/w_train.json; cut -c1-260 gpurun_out/w_train.json
timeout 300 python bench.py --mode train --steps 20 --warmup 3 --coherent 2>/dev/null > gpurun_out/w_train_smooth.json; cut -c1-260 gpurun_out/w_train_smooth.json
timeout 300 python bench.py --mode train --steps 20 --warmup 3 --no-graph 2>/dev/null > gpurun_out/w_train_eager.json; cut -c1-260 gpurun_out/w_train_eager.json
timeout 600 python bench.py > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err; echo "bench exit $?"; cut -c1-250 gpurun_out/w_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/w_bench.json"))
print(d.get("value_with_h2d"), d.get("single_forward_ms"))
PY
