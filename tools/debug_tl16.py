#!/usr/bin/env python3
"""GPU box: where the window-tiled temporal layer differs from the 32 x 32 kernel (debug)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out
ops = HipOps()
F, win = int(os.environ.get("F", 200)), 40
g = torch.Generator().manual_seed(1)
wqkv_kn, wout_kn = torch.randn(64, 768, generator=g) * 0.125, torch.randn(256, 64, generator=g) * 0.0625
wqkv, wout = pack_kn(wqkv_kn).cuda(), pack_kn(wout_kn).cuda()
ws, wo = pack_bf3(wqkv_kn).cuda(), pack_bf3_temporal_out(wout_kn).cuda()
ang = torch.arange(F).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None]
rc, rs, band = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda(), torch.randn(2 * win + 1, 8, generator=g).cuda()
for HW in (64, 256):
    x = torch.randn(F * HW, 64, generator=g).cuda()
    ops.temporal_flags = 4
    a = ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=ws, wout_bf3p=wo)
    ops.temporal_flags = 5
    for rep in range(4):
        b = ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=ws, wout_bf3p=wo)
        d = (a - b).abs().view(F, HW, 64)
        bad = d > 1e-4
        print(f"HW={HW} rep {rep}: max {float(d.max()):.3e}, bad elements {int(bad.sum())}")
        for px in sorted(set(bad.nonzero()[:, 1].tolist())):
            bp = bad[:, px]
            fr = sorted(set(bp.nonzero()[:, 0].tolist()))
            ch = sorted(set(bp.nonzero()[:, 1].tolist()))
            per_group = [float(d[fr[0]:fr[-1] + 1, px, 16 * c:16 * c + 16].max()) for c in range(4)]
            print(f"   pixel {px}: frames {fr[0]}..{fr[-1]} ({len(fr)}), channels {len(ch)} [{ch[0]}..{ch[-1]}], max err per 16-channel group {['%.1e' % v for v in per_group]}")
