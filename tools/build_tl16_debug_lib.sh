#!/bin/bash
# EXPERIMENTAL library (not shipped): temporal_layer16.hip compiled with extra -D flags (DAWN_TL16_DUMP: per (pixel, head, tile) softmax
# statistics and O^T to a debug buffer; DAWN_TL_TIMING: s_memtime stamps).  Usage: tools/build_tl16_debug_lib.sh -DDAWN_TL16_DUMP
set -e
cd "$(dirname "$0")/.."
./build_lib.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c dawn-pytorch_amd/csrc/temporal_layer16.hip -o build/temporal_layer16_debug.o
objs=""
for f in dawn_api conv_gemm conv3x3_wino conv3x3_wino4 ubench pbnet norm temporal_attn temporal_layer temporal_layer16_debug spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do objs="$objs build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ubench/libdawn_hip_tl16debug.bin
echo "built tools/ubench/libdawn_hip_tl16debug.bin"
