#!/bin/bash
# Builds the kernel library (gfx950) in-tree, in its two storage variants (common_hip.h):
#   lib/libmammoclip_hip.so      bf16 storage / operands (default)
#   lib/libmammoclip_hip_f16.so  IEEE f16 storage / operands (-DMC_F16; opt-in: MC_STORAGE=f16)
# hipcc cross-compiles without a GPU.  MC_BUILD_F16=0 skips the second variant.  All stale objects of both variants are
# compiled concurrently (the longest translation unit bounds the wall time), then the two libraries are linked.
# MC_REBUILD=1: clean build -- every object of both variants is recompiled from source (the make-like default only
# recompiles objects older than their sources; prebuilt objects travel with the tree).
set -e
cd "$(dirname "$0")"
if [ "${MC_REBUILD:-0}" = "1" ]; then rm -f ../lib/*.o ../lib/f16/*.o ../lib/libmammoclip_hip.so ../lib/libmammoclip_hip_f16.so; fi
SRCS="gemm gemm256 gemm256_tn fp8 gemm_rows gemm_wgrad_rows conv conv_lane bnact bnfold bert attn head optim util"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
pids=()
compile_variant() {   # $1 = object directory, $2 = extra flags
  local OUT=$1 EXTRA=$2
  mkdir -p $OUT
  for f in $SRCS; do
    if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ] || [ common_hip.h -nt $OUT/$f.o ] || [ ../../include/mammoclip_hip.h -nt $OUT/$f.o ]; then
      ( hipcc $FLAGS $EXTRA -c $f.hip -o $OUT/$f.o.tmp && mv $OUT/$f.o.tmp $OUT/$f.o ) &
      pids+=($!)
    fi
  done
}
link_variant() {      # $1 = object directory, $2 = library
  local objs=""
  for f in $SRCS; do objs="$objs $1/$f.o"; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o $2 $objs
  echo "built $2"
}
compile_variant ../lib ""
if [ "${MC_BUILD_F16:-1}" != "0" ]; then compile_variant ../lib/f16 "-DMC_F16"; fi
for p in "${pids[@]}"; do wait $p; done
link_variant ../lib ../lib/libmammoclip_hip.so
if [ "${MC_BUILD_F16:-1}" != "0" ]; then link_variant ../lib/f16 ../lib/libmammoclip_hip_f16.so; fi
