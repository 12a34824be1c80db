#!/usr/bin/env python3
"""GPU microbenchmark of the LayerNorm row-statistics pass (dawn_ln_rowstats) at the benchmark's shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
ops = HipOps()
for rows, C0, C1 in ((819200, 64, 64), (204800, 128, 0), (204800, 128, 128), (51200, 256, 0), (51200, 256, 256), (12800, 512, 512)):
    x = torch.randn(rows, C0, device="cuda"); x2 = torch.randn(rows, C1, device="cuda") if C1 else None
    ops.ln_rowstats(x, x2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.ln_rowstats(x, x2)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"ln_rowstats rows={rows} C={C0}+{C1}: {us:7.1f} us  {rows * (C0 + C1) * 4 / us / 1e6:5.2f} TB/s")
