# Developer helper: run one command under several builds of the library on ONE box: bash scripts/ab_libs.sh "v1 v2 ..." <command...>
# (ab/<v>.so; "base" = the library as built)
L=mammo_clip_amd/lib/libmammoclip_hip.so
cp $L /tmp/keep.so
VS=$1; shift
for v in $VS; do
  if [ $v = base ]; then cp /tmp/keep.so $L; else cp ab/$v.so $L; fi
  echo "=== $v"
  "$@"
done
cp /tmp/keep.so $L
