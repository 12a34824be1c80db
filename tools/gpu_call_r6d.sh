#!/bin/bash
# Round 6, call D: the whole GPU suite with the window-tiled temporal layer as the default.
cd "$(dirname "$0")/.."
O=gpurun_out/r6d; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
