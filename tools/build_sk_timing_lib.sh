#!/bin/bash
# Build build/libdawn_hip_sktiming.so: the normal library with the s_memtime-instrumented stream-K 3x3 conv kernel
# (-DDAWN_ABLATION on conv3x3_sk.hip only).  Loaded by tools/conv_sk_phase_timing.py through DAWN_HIP_LIB; the shipped
# dawn-pytorch_amd/libdawn_hip.so is not touched.
set -e
cd "$(dirname "$0")/.."
./build_lib.sh > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_ABLATION -c dawn-pytorch_amd/csrc/conv3x3_sk.hip -o build/conv3x3_sk_timing.o
objs=""
for f in dawn_api conv_gemm conv3x3_sk_timing ubench pbnet norm temporal_attn temporal_layer spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do objs="$objs build/$f.o"; done
mkdir -p tools/ubench
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/ubench/libdawn_hip_sktiming.bin
echo "built tools/ubench/libdawn_hip_sktiming.bin"
