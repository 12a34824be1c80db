#!/usr/bin/env python3
"""GPU: the unfused levels' temporal attention core (dawn_temporal_attn_ex) at the benchmark shapes -- the split-operand (bf16 pipe)
kernel vs the fp32-MFMA kernel (flags = 1), alternating."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps

ops = HipOps()
dev = "cuda"
F, win = 200, 40
ang = torch.arange(F + 2 * win).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None]
rc, rs = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
band = (torch.randn(2 * win + 1, 8) * 0.1).to(dev)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for HW, Fext, q0, Fq in ((1024, 200, 0, 200), (256, 200, 0, 200), (64, 200, 0, 200), (1024, 280, 40, 200)):
    qkv = torch.randn(Fext * HW, 768, device=dev)
    res = []
    for r in range(2):
        for fl in (0, 1):
            ops.temporal_attn_flags = fl
            res.append(timeit(lambda: ops.temporal_attn(qkv, Fext, HW, q0, Fq, win, rc, rs, band)))
    ops.temporal_attn_flags = 0
    a = ops.temporal_attn(qkv, Fext, HW, q0, Fq, win, rc, rs, band)
    ops.temporal_attn_flags = 1
    b = ops.temporal_attn(qkv, Fext, HW, q0, Fq, win, rc, rs, band)
    ops.temporal_attn_flags = 0
    print(f"HW={HW:5d} Fext={Fext} q0={q0} Fq={Fq}: split {res[0]:7.1f} {res[2]:7.1f} us   fp32 {res[1]:7.1f} {res[3]:7.1f} us   max |diff| {float((a - b).abs().max()):.2e}")
