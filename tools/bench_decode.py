#!/usr/bin/env python3
"""GPU box: throughput of the LFG flow decode -- `FlowDecoder.decode_clip` on a synthetic clip with
the shipped generator topology (64/128/256 channels, 6 bottleneck blocks) and seeded random weights.

    python tools/bench_decode.py [--res 256] [--frames 200] [--iters 3] [--cpu-frames 2] [--chunk 64]

Prints one JSON line: decoded frames/s, algorithmic TFLOP/s of the convolutions (dense math of GEN:138-171 per frame,
encoder excluded -- it runs once per clip here and once per FRAME in the reference), the per-kernel-class time split
measured with HIP events, and the CPU oracle (oracle/lfg_ref.py) timed on `--cpu-frames` frames beside it."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def lfg_state_dict(seed=0, be=64, max_features=512, n_down=2, n_bott=6):
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(p, co, ci, k):
        sd[p + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (ci * k * k) ** -0.5
        sd[p + ".bias"] = torch.randn(co, generator=g) * 0.1

    def bn(p, c):
        sd[p + ".weight"] = 1 + 0.2 * torch.randn(c, generator=g)
        sd[p + ".bias"] = 0.2 * torch.randn(c, generator=g)
        sd[p + ".running_mean"] = 0.2 * torch.randn(c, generator=g)
        sd[p + ".running_var"] = torch.rand(c, generator=g) + 0.5

    conv("first.conv", be, 3, 7); bn("first.norm", be)
    for i in range(n_down):
        ci, co = min(max_features, be * 2 ** i), min(max_features, be * 2 ** (i + 1))
        conv(f"down_blocks.{i}.conv", co, ci, 3); bn(f"down_blocks.{i}.norm", co)
    for i in range(n_down):
        ci, co = min(max_features, be * 2 ** (n_down - i)), min(max_features, be * 2 ** (n_down - i - 1))
        conv(f"up_blocks.{i}.conv", co, ci, 3); bn(f"up_blocks.{i}.norm", co)
    cb = min(max_features, be * 2 ** n_down)
    for i in range(n_bott):
        for j in (1, 2):
            conv(f"bottleneck.r{i}.conv{j}", cb, cb, 3); bn(f"bottleneck.r{i}.norm{j}", cb)
    sd["final.weight"] = torch.randn(3, be, 7, 7, generator=g) * (be * 49) ** -0.5
    sd["final.bias"] = torch.randn(3, generator=g) * 0.1
    return sd


def decode_flops_per_frame(res, be=64, n_bott=6):
    """Dense multiply-adds x2 of the per-frame part of GEN:138-171 (bottleneck, up blocks, final conv)."""
    hb = res // 4
    cb = be * 4
    f = n_bott * 2 * 2.0 * hb * hb * 9 * cb * cb
    f += 2.0 * (2 * hb) ** 2 * 9 * cb * (cb // 2)
    f += 2.0 * (4 * hb) ** 2 * 9 * (cb // 2) * be
    f += 2.0 * res * res * 49 * be * 3
    return f


def synthetic_motion(T, h, device, seed=123):
    g = torch.Generator().manual_seed(seed)
    lin = (torch.arange(h, dtype=torch.float32) + 0.5) / h * 2 - 1
    yy, xx = torch.meshgrid(lin, lin, indexing="ij")
    grid = torch.stack((xx, yy), 0).view(1, 2, 1, h, h) + torch.randn(1, 2, T, h, h, generator=g) * 0.1
    conf = torch.rand(1, 1, T, h, h, generator=g)
    return grid.to(device), conf.to(device)


def run(res=256, frames=200, iters=3, chunk=64, cpu_frames=2, seed=0):
    from dawn_pytorch_amd.flow_decoder import FlowDecoder
    from dawn_pytorch_amd.ops import HipOps
    dev = torch.device("cuda:0")
    sd = lfg_state_dict(seed)
    ops = HipOps()
    dec = FlowDecoder(sd, dev, ops=ops, chunk=chunk)
    img = torch.rand(1, 3, res, res, generator=torch.Generator().manual_seed(1)).to(dev)
    grid, conf = synthetic_motion(frames, res // 4, dev)
    dec.decode_clip(img, grid, conf)                       # warm-up (weights resident, attributes set)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = dec.decode_clip(img, grid, conf)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    fl = decode_flops_per_frame(res) * frames
    # per-class split with HIP events around every op of one more decode
    classes = {}
    real = {}
    for name in ("conv_gemm", "warp_blend", "affine_act", "final_conv_blend", "init_conv_x", "bn_relu_pool2"):
        fn = getattr(ops, name)
        real[name] = fn

        def wrap(*a, _fn=fn, _name=name, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = _fn(*a, **k)
            e1.record()
            classes.setdefault(_name, []).append((e0, e1))
            return r
        setattr(ops, name, wrap)
    dec.decode_clip(img, grid, conf)
    torch.cuda.synchronize()
    for name, fn in real.items():
        setattr(ops, name, fn)
    split = {k: {"launches": len(v), "ms": sum(a.elapsed_time(b) for a, b in v)} for k, v in classes.items()}
    result = {"metric": "decoded frames/sec (LFG flow decode, FD:372-385 batched)", "value": frames / dt, "unit": "frames/s",
              "config": {"workload": f"{res}x{res}, {frames} frames, generator 64/128/256 ch, 6 bottleneck blocks",
                         "chunk": chunk}, "dtype": "f32 (3x3 convs: exact 3-way bf16 operand split, fp32 accumulate)",
              "data": "synthetic", "ms_per_clip": dt * 1e3, "algorithmic_tflop_per_clip": fl / 1e12,
              "algorithmic_tflops": fl / dt / 1e12, "kernel_ms": split, "all_iters_ms": [t * 1e3 for t in ts]}
    if cpu_frames > 0:
        from oracle import lfg_ref
        n = torch.get_num_threads()
        cg, cc = grid[:, :, :cpu_frames].cpu(), conf[:, :, :cpu_frames].cpu()
        t0 = time.perf_counter()
        want = lfg_ref.decode_clip(sd, img.cpu(), cg, cc, chunk=1)
        tc = time.perf_counter() - t0
        err = float((out["sample_out_vid"][:, :, :cpu_frames].cpu() - want["sample_out_vid"]).abs().max())
        result["cpu_baseline"] = {"value": cpu_frames / tc, "unit": "frames/s", "cores": n, "kind": "port",
                                  "sample": f"{cpu_frames} frames of the same clip through oracle/lfg_ref.py "
                                            "(per-frame encoder re-run included, as in the reference)"}
        result["max_abs_err_vs_oracle"] = err
    return result


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--cpu-frames", type=int, default=2)
    a = ap.parse_args()
    print(json.dumps(run(a.res, a.frames, a.iters, a.chunk, a.cpu_frames)))
