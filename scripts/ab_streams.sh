# A/B of MC_STREAMS settings on ONE box: bash scripts/ab_streams.sh "0 1 3" [bench args]
VS=$1; shift
for r in 1 2; do for v in $VS; do
  MC_STREAMS=$v python bench.py "$@" --no-cpu-baseline --no-n8-load 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MC_STREAMS=$v', d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))"
done; done
