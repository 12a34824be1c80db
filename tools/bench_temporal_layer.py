#!/usr/bin/env python3
"""GPU box: the fused 64-channel temporal layer at the benchmark's level-0 shape (200 frames, 4096 pixels), HIP events.   python tools/bench_temporal_layer.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out
ops = HipOps()
F, win = 200, 40
g = torch.Generator().manual_seed(1)
wqkv_kn, wout_kn = torch.randn(64, 768, generator=g) * 0.125, torch.randn(256, 64, generator=g) * 0.0625
wqkv, wout = pack_kn(wqkv_kn).cuda(), pack_kn(wout_kn).cuda()
ws, wo = pack_bf3(wqkv_kn).cuda(), pack_bf3_temporal_out(wout_kn).cuda()
ang = torch.arange(F).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None]
rc, rs, band = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda(), torch.randn(2 * win + 1, 8, generator=g).cuda()
# flags: 4 = the 32 x 32-tile kernel (WMODE 3), 5 = the window-tiled 16-query kernel (WMODE 4, round 6), alternating
for HW in (4096, 1024):
    x = torch.randn(F * HW, 64, generator=g).cuda()
    out = torch.empty_like(x)
    for rep in range(3):
        for flags, name in ((4, "wmode3 32x32 tiles"), (5, "wmode4 window-tiled"), (6, "wmode5 tile per wave")):
            ops.temporal_flags = flags
            for _ in range(2):
                ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=ws, wout_bf3p=wo, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=ws, wout_bf3p=wo, out=out)
            e1.record()
            torch.cuda.synchronize()
            print(f"temporal_layer_c64 F={F} HW={HW} {name}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch")
# BASELINE configs[1]: a 120-query segment on a 200-row window (128 x 128: 1024 pixel columns)
HW = 1024
x = torch.randn(F * HW, 64, generator=g).cuda()
out = torch.empty(120 * HW, 64, device="cuda")
for flags, name in ((4, "wmode3 32x32 tiles"), (5, "wmode4 window-tiled"), (6, "wmode5 tile per wave")):
    ops.temporal_flags = flags
    for _ in range(2):
        ops.temporal_layer_c64(x, F, HW, 40, 120, win, wqkv, wout, rc, rs, band, wqkv_bf3=ws, wout_bf3p=wo, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.temporal_layer_c64(x, F, HW, 40, 120, win, wqkv, wout, rc, rs, band, wqkv_bf3=ws, wout_bf3p=wo, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"temporal_layer_c64 segment Fext=200 q0=40 Fq=120 HW={HW} {name}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per launch")
