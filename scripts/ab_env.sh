# A/B of environment settings on ONE box: bash scripts/ab_env.sh "VAR=a VAR=b ..." [bench args]   (two alternating rounds)
VS=$1; shift
for r in 1 2; do for v in $VS; do
  env $v python bench.py "$@" --no-cpu-baseline --no-n8-load --roofline-in-timed-region 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done; done
