#!/bin/bash
# Build a variant of libdawn_hip.so for same-box A/B runs (DAWN_HIP_LIB=<out>): ONE source file recompiled with extra flags, the rest
# of the objects taken from build/ (run ./build_lib.sh first).
#   tools/build_variant_lib.sh sla_layer tools/ubench/libdawn_hip_sla_fp32out.bin -DDAWN_SLA_OUT_FP32
set -e
cd "$(dirname "$0")/.."
f=$1; out=$2; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c dawn-pytorch_amd/csrc/$f.hip -o build/${f}_variant.o
objs=$(ls build/*.o | grep -v "_variant.o" | grep -v "build/$f.o" | tr '\n' ' ')
hipcc --offload-arch=gfx950 -shared -fPIC $objs build/${f}_variant.o -o $out
rm -f build/${f}_variant.o
echo "built $out"
