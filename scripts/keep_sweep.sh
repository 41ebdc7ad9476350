for cfg in "2 7" "3 11" "3 12" "2 8" "4 11" "2 7"; do set -- $cfg
 timeout 300 python bench.py --keep-mode $1 --keep-kept $2 --steps 2 --warmup 1 --no-cpu-baseline --no-n8-load 2>&1 | tail -1 | python -c "
import sys,json
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print('mode=$1 keep=$2', d['ms_per_step'], d['value'], d['config'].get('peak_hbm_gb'))
except Exception as e: print('mode=$1 keep=$2 FAILED', l[-300:])
"
done
