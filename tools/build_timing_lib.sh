#!/bin/bash
# Rebuild libdawn_hip.so with the s_memtime-instrumented temporal layer kernel (run the normal build_lib.sh afterwards).
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_TL_TIMING -c dawn-pytorch_amd/csrc/temporal_layer.hip -o build/temporal_layer.o
# conv ablation / s_memtime kernels + dawn_conv_set_debug exist only in this build (-DDAWN_ABLATION)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDAWN_ABLATION -c dawn-pytorch_amd/csrc/conv_gemm.hip -o build/conv_gemm.o
objs=""
for f in dawn_api conv_gemm norm temporal_attn temporal_layer spatial_attn sla_layer cond_xattn xattn_layer misc sampler flow_decode dawn_ctx hubert; do objs="$objs build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o dawn-pytorch_amd/libdawn_hip.so
touch dawn-pytorch_amd/csrc/temporal_layer.hip dawn-pytorch_amd/csrc/conv_gemm.hip   # force a clean rebuild of this object by build_lib.sh
echo "built instrumented dawn-pytorch_amd/libdawn_hip.so"
