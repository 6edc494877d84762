set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/conf
export TMPDIR=/tmp
for f in tests/test_gpu_kernels.py tests/test_gpu_model.py; do
  timeout 900 python -m pytest $f -m gpu -q -x --timeout=600 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
done
python scripts/layer_table.py 512 640 5 > gpurun_out/conf/layer_table.txt 2>&1; grep -n "winograd \|lds " gpurun_out/conf/layer_table.txt | head -20; tail -2 gpurun_out/conf/layer_table.txt
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-train --no-other-configs --no-stream-inputs --no-coherent 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['single_forward_ms'])"; done
