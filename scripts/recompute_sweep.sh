# keep_graphs x recompute-mode sweep at the N = 8 per-GPU load of the default workload (128 pairs = 4 micro-batches)
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -k "recompute" 2>&1 | tail -3
for cfg in "$@"; do set -- $cfg
 python bench.py --batch $3 --micro-batches $(( $3 / 32 )) --keep-graphs $2 --recompute $1 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('pairs=$3 rc=$1 keep=$2', d['ms_per_step'], d['value'], d['config']['peak_hbm_gb'])
except Exception as e: print('pairs=$3 rc=$1 keep=$2 FAILED', l[-300:])
"
done
