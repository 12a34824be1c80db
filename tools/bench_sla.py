#!/usr/bin/env python3
"""GPU microbenchmark of the unfused linear-attention core (dawn_sla_context + dawn_sla_apply) at the 128/256-channel levels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
ops = HipOps()
for F, HW in ((200, 1024), (200, 256), (200, 64)):
    qkv = torch.randn(F * HW, 768, device="cuda")
    ops.sla(qkv, F, HW); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.sla(qkv, F, HW)
    e1.record(); torch.cuda.synchronize()
    print(f"sla F={F} HW={HW}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us (context + apply)")
