#!/bin/bash
# Build libdawn_hip.so for gfx950 (cross-compiles without a GPU).  In-tree output so it ships with gpurun.
set -e
cd "$(dirname "$0")"
SRC="dawn-pytorch_amd/csrc"
OUT="dawn-pytorch_amd/libdawn_hip.so"
mkdir -p build
objs=""
for f in dawn_api conv_gemm conv3x3_wino conv3x3_wino4 ubench pbnet norm temporal_attn temporal_layer temporal_layer16 spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do
  if [ ! -f build/$f.o ] || [ $SRC/$f.hip -nt build/$f.o ] || [ $SRC/dawn_common.h -nt build/$f.o ] || [ $SRC/temporal_layer16.h -nt build/$f.o ] || [ include/dawn_hip.h -nt build/$f.o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $SRC/$f.hip -o build/$f.o &
  fi
  objs="$objs build/$f.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $OUT"
