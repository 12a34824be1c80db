#!/usr/bin/env python3
"""GPU microbenchmark of the temporal attention core of the unfused levels (dawn_temporal_attn_ex) at the benchmark's shapes:
    python tools/bench_temporal_attn.py     -> us per launch for flags 0 (automatic: the 32 x 32 split kernel from 128 pixel columns), 4 (the 13-wave kernel), 1 (fp32 MFMA)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
ops = HipOps()
dev = "cuda"
for (F, HW, q0, Fq) in ((200, 1024, 0, 200), (200, 256, 0, 200), (200, 64, 0, 200), (200, 1024, 40, 120)):
    torch.manual_seed(0)
    qkv = torch.randn(F * HW, 768, device=dev)
    ang = torch.arange(F).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None]
    rc, rs = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    band = (torch.randn(81, 8) * 0.1).to(dev)
    qkv_ph = qkv.view(F, HW, 3, 8, 32).permute(1, 3, 2, 0, 4).contiguous().view(F * HW, 768)      # [pixel][head][q|k|v][row][32]
    for flags in (0, 4, 4 | 16, 1):
        ops.temporal_attn_flags = flags
        src = qkv_ph if flags & 16 else qkv
        for _ in range(2):
            o = ops.temporal_attn(src, F, HW, q0, Fq, 40, rc, rs, band)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            o = ops.temporal_attn(src, F, HW, q0, Fq, 40, rc, rs, band)
        e1.record()
        torch.cuda.synchronize()
        print(f"F={F} HW={HW} q0={q0} Fq={Fq} flags={flags}: {e0.elapsed_time(e1) * 100:8.1f} us   checksum {float(o.double().sum()):.6f}")
ops.temporal_attn_flags = 0
