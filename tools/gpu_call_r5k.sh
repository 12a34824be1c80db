#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5k; mkdir -p $O
timeout 900 python tools/max_clip_length.py --try-frames 62000 2>&1 | grep -v amdgpu | tee $O/max_clip_length.log | tail -5
