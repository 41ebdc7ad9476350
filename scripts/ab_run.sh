# A/B of two builds of the library on ONE box: bash scripts/ab_run.sh '<command printing the numbers to compare>'
# expects ab/old.so and ab/new.so; runs the command with each (old, new, old, new)
L=mammo_clip_amd/lib/libmammoclip_hip.so
cp $L /tmp/keep.so
for r in 1 2; do for v in old new; do
  cp ab/$v.so $L
  echo "== $v"
  bash -c "$1"
done; done
cp /tmp/keep.so $L
