#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5j; mkdir -p $O
DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_tltiming.bin timeout 300 python tools/temporal_phase_timing.py 200 2>&1 | grep -v amdgpu | tee $O/temporal_phase_timing.txt | head -30
