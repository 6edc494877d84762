#!/bin/bash
# round 3, call a: LDS-DMA semantics probe, conv_lds phase timeline (probe build), baseline bench of this box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export TMPDIR=/tmp
./scripts/probes/lds_dma_probe > gpurun_out/lds_dma_probe.txt 2>&1; cat gpurun_out/lds_dma_probe.txt
MVSTER_LIB=$PWD/mvster_amd/csrc/libmvster_hip_tl.so timeout 300 python scripts/conv_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_timeline.txt; cat gpurun_out/conv_timeline.txt
timeout 300 python bench.py --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/bench_a.json; cat gpurun_out/bench_a.json | cut -c1-400
