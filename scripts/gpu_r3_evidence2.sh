#!/bin/bash
# the evidence run, preceded by the tests of whatever changed last (fail fast: nothing else runs if they do not pass)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "${QUICK_K:-stage_terms or losses_vs_reference or sinkhorn}" > gpurun_out/quick_tests.txt 2>&1
rc=$?
grep -E "passed|failed|error" gpurun_out/quick_tests.txt | tail -3
if [ $rc -ne 0 ]; then tail -40 gpurun_out/quick_tests.txt; echo "QUICK TESTS FAILED"; exit 1; fi
bash scripts/gpu_r3_evidence.sh
