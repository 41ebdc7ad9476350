# A/B of two builds of the library on ONE box (box-to-box variance is ~1 %): bash scripts/ab_bench.sh [bench args]
# expects ab/old.so and ab/new.so; alternates old/new twice and prints ms_per_step
L=mammo_clip_amd/lib/libmammoclip_hip.so
cp $L /tmp/keep.so
for r in 1 2; do for v in old new; do
  cp ab/$v.so $L
  python bench.py "$@" --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'])"
done; done
cp /tmp/keep.so $L
