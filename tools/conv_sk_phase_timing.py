#!/usr/bin/env python3
"""GPU tool: per-phase s_memtime profile of the persistent stream-K 3x3 conv kernel (instrumented build:
tools/build_sk_timing_lib.sh, loaded through DAWN_HIP_LIB).
    DAWN_HIP_LIB=tools/ubench/libdawn_hip_sktiming.bin python tools/conv_sk_phase_timing.py [--C1 64] [--leave 8]
Wave 0 of every workgroup stamps units 2..5 of its range (TSTAMP() in conv3x3_sk.hip): per unit
  top | [requests issued, MFMAs issued, loads landed, barrier passed] x 3 kernel rows | planes written.
Prints the mean / p10 / p90 cycles between consecutive stamps over the workgroups."""
import argparse, ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pack import pack_kn, pack_bf3

ap = argparse.ArgumentParser()
ap.add_argument("--C0", type=int, default=64)
ap.add_argument("--C1", type=int, default=0)
ap.add_argument("--N", type=int, default=64)
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--policy", type=lambda v: int(v, 0), default=0x5C0D)
ap.add_argument("--leave", type=int, default=0, help="policy bits 20..23: grid = (16 - n)/16 of the resident slots")
a = ap.parse_args()
ops = HipOps()
F, H, W, C0, C1, N = a.frames, 64, 64, a.C0, a.C1, a.N
rows, K = F * H * W, 9 * (C0 + C1)
torch.manual_seed(0)
x0 = torch.randn(rows, C0, device="cuda")
x1 = torch.randn(rows, C1, device="cuda") if C1 else None
w_kn = torch.randn(K, N) * K ** -0.5
w, ws = pack_kn(w_kn).cuda(), pack_bf3(w_kn).cuda()
dbg = torch.zeros(1024 * 64, dtype=torch.int64, device="cuda")
ops.L.dawn_conv_sk_set_debug.argtypes = [ctypes.c_void_p]
assert ops.L.dawn_conv_sk_set_debug(dbg.data_ptr()) == 0
# hand-off scratch of the stream-K kernel (experimental library only: the shipped one has neither these entry points nor the kernel)
ops.L.dawn_conv_sk_workspace_bytes.restype = ctypes.c_size_t
ops.L.dawn_conv_sk_workspace_init.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
ops.sk_ws = torch.empty(int(ops.L.dawn_conv_sk_workspace_bytes()), dtype=torch.uint8, device="cuda")
assert ops.L.dawn_conv_sk_workspace_init(ops.sk_ws.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
part = ops.conv_gn_part(rows, N, x0)
ops.conv_policy = a.policy | (a.leave << 20)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    dbg.zero_()
    e0.record()
    out = ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, w_bf3=ws, gn_part=part)
    e1.record()
    torch.cuda.synchronize()
t = dbg.cpu().numpy().reshape(1024, 64)
t = t[t[:, 0] != 0]
span = t[:, 62:64].astype(np.float64)
t = t[:, :62]
nst = int((t[0] != 0).sum())
t = t[:, :nst].astype(np.float64)
t0 = span[:, 0].min()
dur = span[:, 1] - span[:, 0]
print(f"workgroup start (rel): p0 {np.percentile(span[:, 0] - t0, 0):.0f} p50 {np.percentile(span[:, 0] - t0, 50):.0f} p100 {np.percentile(span[:, 0] - t0, 100):.0f};  "
      f"end (rel): p0 {np.percentile(span[:, 1] - t0, 0):.0f} p10 {np.percentile(span[:, 1] - t0, 10):.0f} p50 {np.percentile(span[:, 1] - t0, 50):.0f} "
      f"p90 {np.percentile(span[:, 1] - t0, 90):.0f} p100 {np.percentile(span[:, 1] - t0, 100):.0f};  duration: min {dur.min():.0f} mean {dur.mean():.0f} max {dur.max():.0f}")
print(f"C0+C1 = {C0}+{C1}, grid {t.shape[0]} workgroups, {nst} stamps each; launch {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented)")
names = ["top"] + [f"ky{k}:{n}" for k in range(3) for n in ("requests issued", "MFMAs issued", "loads landed", "barrier passed")] + ["planes written"]
d = np.diff(t, axis=1)
per = len(names)
for i in range(nst - 1):
    print(f"  {i:2d}->{i+1:2d} {names[i % per]:>22s} -> {names[(i + 1) % per]:<22s}: {d[:, i].mean():8.0f} {np.percentile(d[:, i], 10):8.0f} {np.percentile(d[:, i], 90):8.0f}")
unit = (t[:, per:2 * per] - t[:, :per]).mean(axis=0) if nst >= 2 * per else None
if unit is not None:
    print(f"cycles per unit (stamp k of unit 3 - stamp k of unit 2), mean over stamps and workgroups: {unit.mean():.0f}")
