#!/bin/bash
# Developer helper: build ab/<name>.so = the library with ONE translation unit recompiled under extra flags (or from an edited
# copy of the source): bash scripts/ab_variant.sh <name> <unit> [extra hipcc flags...] ; the other objects come from lib/*.o
# (run csrc/build.sh first).  With SRC=<file> the unit is compiled from that file instead of csrc/<unit>.hip.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; UNIT=$2; shift 2
mkdir -p $ROOT/ab
SRCFILE=${SRC:-$ROOT/mammo_clip_amd/csrc/$UNIT.hip}
TMP=$ROOT/mammo_clip_amd/csrc/_ab_$UNIT.hip
cp $SRCFILE $TMP
trap "rm -f $TMP" EXIT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result "$@" -c $TMP -o $ROOT/ab/${NAME}_$UNIT.o
objs=""
for f in gemm gemm256 gemm256_tn fp8 gemm_rows gemm_wgrad_rows conv conv_lane bnact bnfold bert attn head optim util; do
  if [ $f = $UNIT ]; then objs="$objs $ROOT/ab/${NAME}_$UNIT.o"; else objs="$objs $ROOT/mammo_clip_amd/lib/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/ab/$NAME.so $objs
echo "built ab/$NAME.so"
