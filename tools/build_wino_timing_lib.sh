#!/bin/bash
# EXPERIMENTAL library (not shipped): the normal objects with conv3x3_wino.hip compiled -DDAWN_ABLATION (s_memtime stamps / perf ablations
# selected by the DAWN_WINO_ABL environment variable: 64 = stamps of thread 0 written over the output, see tools/bench_wino.py --stamps).
set -e
cd "$(dirname "$0")/.."
./build_lib.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_ABLATION -c dawn-pytorch_amd/csrc/conv3x3_wino.hip -o build/conv3x3_wino_timing.o
objs=""
for f in dawn_api conv_gemm conv3x3_wino_timing conv3x3_wino4 ubench pbnet norm temporal_attn temporal_layer spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do objs="$objs build/$f.o"; done
mkdir -p tools/ubench
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ubench/libdawn_hip_winotiming.bin
echo "built tools/ubench/libdawn_hip_winotiming.bin"
