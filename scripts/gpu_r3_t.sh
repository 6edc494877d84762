for k in 2 3 4; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-coherent --inflight $k 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('inflight', $k, d['value'], d['ms_per_step'])"; done
