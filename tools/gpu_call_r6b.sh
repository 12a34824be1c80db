#!/bin/bash
# Round 6, call B: window-tiled temporal layer: run-to-run determinism, parity, isolated A/B, phase timing.
cd "$(dirname "$0")/.."
O=gpurun_out/r6b; mkdir -p $O
python tools/debug_tl16.py 2>&1 | grep -v amdgpu | tee $O/determinism.txt
F=184 python tools/debug_tl16.py 2>&1 | grep -v amdgpu | tee -a $O/determinism.txt
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "temporal_layer or temporal_attention_trained or c64_attention_layers_in_place" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python tools/bench_temporal_layer.py 2>&1 | grep -v amdgpu | tee $O/bench_temporal_layer.txt
DAWN_HIP_LIB=$PWD/tools/ubench/libdawn_hip_tl16debug.bin timeout 300 python tools/tl16_phase_timing.py 2>&1 | grep -v amdgpu | tee $O/phase_timing.txt
