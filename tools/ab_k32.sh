#!/bin/bash
# GPU box: the 16x16x32 form of the split 3x3 kernel (policy bit 0x1000000) vs the 32x32x16 form: layout microtest, the conv tests,
# isolated launches (alternating), then the whole benchmark alternating inside this one call.
cd "$(dirname "$0")/.."
O=gpurun_out; mkdir -p $O
tools/ubench/mfma16_layout.bin | tee $O/r3_k32.txt
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "conv" 2>&1 | tail -3 | tee -a $O/r3_k32.txt
for r in 1 2; do
  python tools/bench_conv.py --cases l0_3x3,l0_3x3_cat,l1_3x3,l2_3x3,l3_3x3 --iters 20 --gn --variants 0x580D,0x100580D 2>/dev/null | tee -a $O/r3_k32.txt
done
bash tools/ab_policy.sh "0x580D 0x100580D" 3 | tee -a $O/r3_k32.txt
