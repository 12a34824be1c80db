#!/bin/bash
# EXPERIMENTAL library (not shipped): the normal objects + the persistent stream-K 3x3 conv kernel of round 3
# (tools/ubench/conv3x3_sk.hip, s_memtime-instrumented with -DDAWN_ABLATION) and a conv_gemm.o compiled with
# -DDAWN_WITH_STREAMK so that policy bit 0x400 + dawn_conv_desc.sk_ws reach it.  Loaded by tools/conv_sk_phase_timing.py /
# tools/bench_conv.py through DAWN_HIP_LIB; the shipped dawn-pytorch_amd/libdawn_hip.so is not touched.
set -e
cd "$(dirname "$0")/.."
./build_lib.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_ABLATION -c tools/ubench/conv3x3_sk.hip -o build/conv3x3_sk_timing.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_WITH_STREAMK -c dawn-pytorch_amd/csrc/conv_gemm.hip -o build/conv_gemm_streamk.o
objs=""
for f in dawn_api conv_gemm_streamk conv3x3_sk_timing conv3x3_wino ubench pbnet norm temporal_attn temporal_layer spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do objs="$objs build/$f.o"; done
mkdir -p tools/ubench
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ubench/libdawn_hip_sktiming.bin
echo "built tools/ubench/libdawn_hip_sktiming.bin"
