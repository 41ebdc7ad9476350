#!/bin/bash
# Builds libmammoclip_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
pids=()
for f in gemm gemm256 gemm256_tn fp8 gemm_rows gemm_wgrad_rows conv conv_lane bnact bnfold bert attn head optim util; do
  if [ ! -f $OUT/$f.o ] || [ $f.hip -nt $OUT/$f.o ] || [ common_hip.h -nt $OUT/$f.o ] || [ ../../include/mammoclip_hip.h -nt $OUT/$f.o ]; then
    ( hipcc $FLAGS -c $f.hip -o $OUT/$f.o ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmammoclip_hip.so $OUT/gemm.o $OUT/gemm256.o $OUT/gemm256_tn.o $OUT/fp8.o $OUT/gemm_rows.o $OUT/gemm_wgrad_rows.o $OUT/conv.o $OUT/conv_lane.o $OUT/bnact.o $OUT/bnfold.o $OUT/bert.o $OUT/attn.o $OUT/head.o $OUT/optim.o $OUT/util.o
echo "built $OUT/libmammoclip_hip.so"
