#!/bin/bash
# Builds the kernel library (gfx950) in-tree, in its two storage variants (common_hip.h):
#   lib/libmammoclip_hip.so      bf16 storage / operands (default)
#   lib/libmammoclip_hip_f16.so  IEEE f16 storage / operands (-DMC_F16; opt-in: MC_STORAGE=f16)
# hipcc cross-compiles without a GPU.  MC_BUILD_F16=0 skips the second variant.
set -e
cd "$(dirname "$0")"
SRCS="gemm gemm256 gemm256_tn fp8 gemm_rows gemm_wgrad_rows conv conv_lane bnact bnfold bert attn head optim util"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
build_variant() {   # $1 = object directory, $2 = library, $3 = extra flags
  local OUT=$1 LIBF=$2 EXTRA=$3
  mkdir -p $OUT
  local pids=() objs=""
  for f in $SRCS; do
    objs="$objs $OUT/$f.o"
    if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ] || [ common_hip.h -nt $OUT/$f.o ] || [ ../../include/mammoclip_hip.h -nt $OUT/$f.o ]; then
      ( hipcc $FLAGS $EXTRA -c $f.hip -o $OUT/$f.o ) &
      pids+=($!)
    fi
  done
  for p in "${pids[@]}"; do wait $p; done
  hipcc --offload-arch=gfx950 -shared -fPIC -o $LIBF $objs
  echo "built $LIBF"
}
build_variant ../lib ../lib/libmammoclip_hip.so ""
if [ "${MC_BUILD_F16:-1}" != "0" ]; then
  build_variant ../lib/f16 ../lib/libmammoclip_hip_f16.so "-DMC_F16"
fi
