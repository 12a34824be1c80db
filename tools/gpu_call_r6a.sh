#!/bin/bash
# Round 6, call A: the window-tiled fused temporal layer (temporal_layer16.hip): parity, isolated A/B against the 32 x 32 kernel.
cd "$(dirname "$0")/.."
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "temporal_layer or temporal_attention_trained or c64_attention_layers_in_place" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 300 python tools/bench_temporal_layer.py 2>&1 | grep -v amdgpu | tee $O/bench_temporal_layer.txt
