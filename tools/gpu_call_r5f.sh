#!/bin/bash
# Round 5, GPU call F: F(4x4) kernel tests, the whole GPU suite, and the whole-benchmark A/B of the per-shape F(4x4) policy bit.
cd "$(dirname "$0")/.."
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -3 $O/pytest_gpu.log
rm -f gpurun_out/ab_policy.txt
timeout 900 bash tools/ab_policy.sh "0x300580D 0xB00580D" 3 > $O/ab_policy.log 2>&1
cp gpurun_out/ab_policy.txt $O/; cat $O/ab_policy.txt
timeout 300 python tools/profile_conv_shapes.py --policy 0xB00580D 2>&1 | grep -v amdgpu | head -20 > $O/insitu_f4.txt
timeout 300 python tools/profile_conv_shapes.py 2>&1 | grep -v amdgpu | head -20 > $O/insitu_default.txt
head -8 $O/insitu_f4.txt; head -8 $O/insitu_default.txt
