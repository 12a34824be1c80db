#!/usr/bin/env python3
"""GPU tool: s_memtime phase profile of xattn_c64_kernel<64> (needs xattn_layer.hip built with -DDAWN_XA_TIMING:
    hipcc ... -DDAWN_XA_TIMING -c xattn_layer.hip, link to a second .so and point DAWN_HIP_LIB at it).  --split: bf16-pipe to_q."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3, unpack_kn
ops = HipOps()
dev = "cuda"
F, HW = 200, 4096
torch.manual_seed(0)
x = torch.randn(F * HW, 64, device=dev)
wq = pack_kn(torch.randn(64, 192) * 0.125).to(dev)
wo = [pack_kn(torch.randn(64, 64) * 0.125).to(dev) for _ in range(3)]
g3 = (torch.randn(3, 64) * 0.2 + 1).to(dev)
q_scale = (torch.rand(3, 8) + 0.5).to(dev)
kvtab = torch.randn(F, 3, 128, device=dev)
nulltab = torch.randn(3, 16, device=dev)
xtab = ops.xattn_tables(kvtab, nulltab, q_scale, wo, 64)
wqs = pack_bf3(unpack_kn(wq.cpu())).to(dev) if "--split" in sys.argv else None
dbg = torch.zeros(256 * 8 * 24, dtype=torch.int64, device=dev)
ops.L.dawn_xattn_set_debug.argtypes = [ctypes.c_void_p]
assert ops.L.dawn_xattn_set_debug(dbg.data_ptr()) == 0
for _ in range(2):
    dbg.zero_()
    ops.xattn_layer_c64(x, None, HW, wq, wo, g3, q_scale, kvtab, nulltab, xtab=xtab, wq_bf3=wqs)
    torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(256, 8, 24).astype(np.float64)
names = ["tile start", "x + LN"] + sum([[f"b{b} Q0+heads", f"b{b} Q1+heads", f"b{b} out MFMA", f"b{b} LN+acc"] for b in range(3)], []) + ["stored"]
print("mean cycles between stamps of the 2nd tile of every wave (columns = waves 0..7)")
for i in range(len(names) - 1):
    row = []
    for w in range(8):
        a, b = t[:, w, i], t[:, w, i + 1]
        ok = (a != 0) & (b != 0)
        row.append(f"{(b[ok] - a[ok]).mean():7.0f}" if ok.any() else "      -")
    print(f"  {names[i]:14s} -> {names[i+1]:14s}: " + " ".join(row))
tot = t[:, :, len(names) - 1] - t[:, :, 0]
print("tile total:", " ".join(f"{tot[:, w].mean():7.0f}" for w in range(8)))
