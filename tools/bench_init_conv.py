#!/usr/bin/env python3
"""GPU box: the 3-channel 7x7 init conv (init_conv_x_mfma_kernel) at the benchmark's shape, HIP events.   python tools/bench_init_conv.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dawn_pytorch_amd.ops import HipOps
ops = HipOps()
F, h, w, Co = 200, 64, 64, 64
x = torch.randn(3, F, h, w, device="cuda")
w3 = torch.randn(147, Co, device="cuda") * 147 ** -0.5
fea = torch.randn(h * w, Co, device="cuda")
out = torch.empty(F * h * w, Co, device="cuda")
for _ in range(3):
    ops.init_conv_x(x, w3, fea, F, h, w, Co, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.init_conv_x(x, w3, fea, F, h, w, Co, out=out)
e1.record()
torch.cuda.synchronize()
print(f"init_conv_x {F} x {h} x {w} -> {Co}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch")
