#!/bin/bash
# Round 5, call Q: the F(4x4) kernel after the vector-instruction trim (no spills left in its 32-pixel-row variant) at the level-1 shapes: whole
# benchmark with the shipped per-shape gate against "wherever the geometry fits" (policy bit 0x10000000; the level-1 blocks carry F(4x4) images in this tree), and in situ per shape.
cd "$(dirname "$0")/.."
O=gpurun_out/r5q; mkdir -p $O
timeout 900 bash tools/ab_policy.sh "0x2B00580D 0x3B00580D" 3 > $O/ab_policy.log 2>&1; cp gpurun_out/ab_policy.txt $O/; cat $O/ab_policy.txt
timeout 300 python tools/profile_conv_shapes.py --policy 0x2B00580D 2>&1 | grep -v amdgpu | grep "3x3\|forward" > $O/insitu_default.txt
timeout 300 python tools/profile_conv_shapes.py --policy 0x3B00580D 2>&1 | grep -v amdgpu | grep "3x3\|forward" > $O/insitu_f4_everywhere.txt
timeout 300 python tools/profile_conv_shapes.py --policy 0x2B00580D 2>&1 | grep -v amdgpu | grep "3x3\|forward" > $O/insitu_default2.txt
timeout 300 python tools/profile_conv_shapes.py --policy 0x3B00580D 2>&1 | grep -v amdgpu | grep "3x3\|forward" > $O/insitu_f4_everywhere2.txt
paste -d'\n' $O/insitu_default.txt $O/insitu_f4_everywhere.txt | head -40
