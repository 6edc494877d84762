mkdir -p gpurun_out/r5w
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu -k "prologue or upsample_bilinear_multi or merged_launches or graph_cache or pack_images" 2>&1 | tail -5 > gpurun_out/r5w/tests.txt
F="--no-cpu-baseline --no-stream-inputs --no-coherent --no-batched --no-other-configs --no-train --steps 40 --warmup 10"
for rep in 1 2 3; do for sw in "" "--no-merge-launches"; do
python bench.py $F $sw 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge' if '$sw'=='' else 'separate', 'rep $rep value', l['value'], 'single_ms', l.get('single_forward_ms'), 'api', l['value_api_call']['value'])" 
done; done > gpurun_out/r5w/ab.txt 2>&1
cat gpurun_out/r5w/tests.txt gpurun_out/r5w/ab.txt
