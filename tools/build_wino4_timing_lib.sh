#!/bin/bash
# EXPERIMENTAL library (not shipped): the normal objects with conv3x3_wino4.hip compiled -DDAWN_ABLATION (DAWN_WINO4_ABL = 64: s_memtime
# stamps of every wave written over the output, tools/bench_wino.py --stamps4; 1 / 2 / 4 / 6 / 7: no epilogue / transform / patch DMA).
set -e
cd "$(dirname "$0")/.."
./build_lib.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_ABLATION -c dawn-pytorch_amd/csrc/conv3x3_wino4.hip -o build/conv3x3_wino4_timing.o
objs=""
for f in dawn_api conv_gemm conv3x3_wino conv3x3_wino4_timing ubench pbnet norm temporal_attn temporal_layer spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do objs="$objs build/$f.o"; done
mkdir -p tools/ubench
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ubench/libdawn_hip_wino4timing.bin
echo "built tools/ubench/libdawn_hip_wino4timing.bin"
