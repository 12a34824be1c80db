#!/usr/bin/env python3
"""GPU microbenchmark of the unfused temporal attention core (dawn_temporal_attn) at the 128/256/512-channel levels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import rel_pos_bucket
ops = HipOps()
F, win = 200, 40
rc, rs = ops.rotary_tables(torch.rand(16, device="cuda"), F) if hasattr(ops, "rotary_tables") else (torch.rand(F, 16, device="cuda"), torch.rand(F, 16, device="cuda"))
band = torch.randn(2 * win + 1, 8, device="cuda")
for HW in (1024, 256, 64):
    qkv = torch.randn(F * HW, 768, device="cuda")
    ops.temporal_attn(qkv, F, HW, 0, F, win, rc, rs, band); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.temporal_attn(qkv, F, HW, 0, F, win, rc, rs, band)
    e1.record(); torch.cuda.synchronize()
    print(f"temporal_attn F={F} HW={HW}: {e0.elapsed_time(e1) * 1e3 / 10:7.1f} us")
