#!/bin/bash
# EXPERIMENTAL library (not shipped): the normal objects with temporal_layer.hip compiled -DDAWN_TL_TIMING (s_memtime stamps per wave and
# phase; tools/temporal_phase_timing.py through DAWN_HIP_LIB).  The shipped dawn-pytorch_amd/libdawn_hip.so is not touched.
set -e
cd "$(dirname "$0")/.."
./build_lib.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_TL_TIMING -c dawn-pytorch_amd/csrc/temporal_layer.hip -o build/temporal_layer_timing.o
objs=""
for f in dawn_api conv_gemm conv3x3_wino conv3x3_wino4 ubench pbnet norm temporal_attn temporal_layer_timing spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do objs="$objs build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ubench/libdawn_hip_tltiming.bin
echo "built tools/ubench/libdawn_hip_tltiming.bin"
