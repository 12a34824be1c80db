#!/usr/bin/env python3
"""GPU tool: run UNet evaluations at the benchmark shape with a HIP-event pair around every conv_gemm
launch and print achieved TFLOP/s per distinct GEMM shape (what to tune first).
    python tools/profile_conv_shapes.py [--frames 200] [--res 256] [--iters 3]"""
import argparse, os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=200)
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--no-overlap", action="store_true")
ap.add_argument("--policy", type=lambda v: int(v, 0), default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
T, h = a.frames, a.res // 4
unet, diff = bench.build_model(T, h, 50, dev)
fea, bbox, cond = bench.synthetic_inputs(T, h, dev)
ops = unet._ops()
if a.no_overlap:
    ops.overlap = False
ops.conv_policy = a.policy
from dawn_pytorch_amd.unet_forward import unet_forward
P = unet.packed()
cs = unet.build_clip(torch.cat((fea, bbox), 1)[0].contiguous(), cond[0].contiguous())
x = torch.randn(3, T, h, h, device=dev)
unet_forward(ops, P, cs, x, 500)
torch.cuda.synchronize()
ops.prof = []
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(a.iters):
    unet_forward(ops, P, cs, x, 500)
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for fl, e0, e1, key, _ in ops.prof:
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1; d[1] += e0.elapsed_time(e1); d[2] += fl
tot_ms = sum(d[1] for d in agg.values()); tot_fl = sum(d[2] for d in agg.values())
print(f"forward {t0.elapsed_time(t1)/a.iters:.2f} ms; conv_gemm {tot_ms/a.iters:.2f} ms/forward, {tot_fl/tot_ms/1e9:.1f} TFLOP/s overall")
print(f"{'shape':70s} {'n/fwd':>6s} {'ms/fwd':>8s} {'us/call':>9s} {'TF/s':>7s} {'%conv':>6s}")
for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{key:70s} {d[0]//a.iters:6d} {d[1]/a.iters:8.3f} {d[1]/d[0]*1e3:9.1f} {d[2]/d[1]/1e9:7.1f} {d[1]/tot_ms*100:6.1f}")
