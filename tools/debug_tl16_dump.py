#!/usr/bin/env python3
"""GPU box (debug library: tools/build_tl16_debug_lib.sh -DDAWN_TL16_DUMP, DAWN_HIP_LIB=tools/ubench/libdawn_hip_tl16debug.bin):
run the window-tiled layer repeatedly on the same input and report which (pixel, head, tile) intermediates differ between runs."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd import _lib
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, pack_bf3_temporal_out
ops = HipOps()
L = _lib.lib()
F, win, HW = int(os.environ.get("F", 184)), 40, 256
g = torch.Generator().manual_seed(1)
wqkv_kn, wout_kn = torch.randn(64, 768, generator=g) * 0.125, torch.randn(256, 64, generator=g) * 0.0625
wqkv, wout = pack_kn(wqkv_kn).cuda(), pack_kn(wout_kn).cuda()
ws, wo = pack_bf3(wqkv_kn).cuda(), pack_bf3_temporal_out(wout_kn).cuda()
ang = torch.arange(F).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None]
rc, rs, band = ang.cos().contiguous().cuda(), ang.sin().contiguous().cuda(), torch.randn(2 * win + 1, 8, generator=g).cuda()
x = torch.randn(F * HW, 64, generator=g).cuda()
ops.temporal_flags = 5
dumps = []
for rep in range(12):
    d = torch.zeros(HW, 8, 16, 64, 12, device="cuda")
    L.dawn_temporal16_set_dump.argtypes = [ctypes.c_void_p]
    assert L.dawn_temporal16_set_dump(d.data_ptr()) == 0
    ops.temporal_layer_c64(x, F, HW, 0, F, win, wqkv, wout, rc, rs, band, wqkv_bf3=ws, wout_bf3p=wo)
    torch.cuda.synchronize()
    dumps.append(d.cpu())
# majority vote = reference
ref = torch.stack(dumps).median(0).values
names = ["m", "l", "sum(q)", "sum(p)"] + [f"o{i}" for i in range(8)]
for rep, d in enumerate(dumps):
    diff = (d - ref).abs() > 1e-5 * (1 + ref.abs())
    bad = diff.any(-1).any(-1)                     # (pixel, head, tile)
    idx = bad.nonzero().tolist()
    print(f"rep {rep}: {len(idx)} differing (pixel, head, tile)")
    for (px, h, t) in idx[:12]:
        dl = diff[px, h, t]                         # (lane, 12)
        lanes = dl.any(-1).nonzero().flatten().tolist()
        fields = [names[i] for i in dl.any(0).nonzero().flatten().tolist()]
        print(f"   pixel {px} head {h} tile {t}: {len(lanes)} lanes (first {lanes[:8]}), fields {fields}")
        for fi, nm in enumerate(names):
            ll = dl[:, fi].nonzero().flatten().tolist()
            if ll:
                print(f"      {nm}: {len(ll)} lanes {ll[:64]}")
        ql = dl[:, 2].nonzero().flatten().tolist()
        if ql:
            print(f"      sum(q) got {[round(d[px, h, t, q, 2].item(), 4) for q in ql[:16]]}")
            print(f"             ref {[round(ref[px, h, t, q, 2].item(), 4) for q in ql[:16]]}")
        l0 = lanes[0]
        print(f"      lane {l0}: got {[round(v, 4) for v in d[px, h, t, l0].tolist()]}")
        print(f"               ref {[round(v, 4) for v in ref[px, h, t, l0].tolist()]}")
