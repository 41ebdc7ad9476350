# A/B/C... of several builds of the library on ONE box: bash scripts/ab_bench_multi.sh "old new v2" [bench args]
# (ab/<name>.so each; two alternating rounds; prints ms_per_step and the metric value)
L=mammo_clip_amd/lib/libmammoclip_hip.so
cp $L /tmp/keep.so
VS=$1; shift
for r in 1 2; do for v in $VS; do
  cp ab/$v.so $L
  python bench.py "$@" --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done; done
cp /tmp/keep.so $L
