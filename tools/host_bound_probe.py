#!/usr/bin/env python3
"""GPU tool: is the Python-orchestrated evaluation host-bound?  Times the HOST enqueue duration of one UNet evaluation
(no synchronisation) against its GPU duration, at the benchmark shapes; and the CPU-oracle thread-count sweep for
bench.py's cpu_baseline (host cores of the GPU box)."""
import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dawn_pytorch_amd.unet_forward import unet_forward

dev = torch.device("cuda", 0)
res = {}
for (T, r) in ((200, 256), (400, 128)):
    h = r // 4
    unet, diff = bench.build_model(T, h, 50, dev)
    fea, bbox, cond = bench.synthetic_inputs(T, h, dev)
    ops, P = unet._ops(), unet.packed()
    cs = unet.build_clip(torch.cat((fea, bbox), 1)[0].contiguous(), cond[0].contiguous())
    x = torch.randn(3, T, h, h, device=dev)
    for overlap in (True, False):
        ops.overlap = overlap
        for _ in range(2):
            unet_forward(ops, P, cs, x, 500)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            unet_forward(ops, P, cs, x, 500)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[f"T{T}_{r}px_overlap{int(overlap)}"] = {"host_enqueue_ms_per_eval": (t1 - t0) / n * 1e3, "total_ms_per_eval": (t2 - t0) / n * 1e3}
        print(f"T={T} {r}px overlap={overlap}: host enqueue {(t1 - t0) / n * 1e3:.2f} ms / eval, wall {(t2 - t0) / n * 1e3:.2f} ms / eval", flush=True)
    del unet, diff, cs
    torch.cuda.empty_cache()

if "--cpu-sweep" in sys.argv:
    from oracle import dawn_oracle as O
    unet, _ = bench.build_model(8, 64, 50, "cpu")
    sd = {"denoise_fn." + k: v.detach() for k, v in unet.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    Ts, h = 24, 64
    fea = torch.randn(1, 272, h, h, generator=g); cond = torch.randn(1, Ts, 1032, generator=g); x = torch.randn(1, 3, Ts, h, h, generator=g)
    xin = torch.cat((x, fea.unsqueeze(2).expand(-1, -1, Ts, -1, -1)), 1)
    with torch.no_grad():
        O.unet_forward(sd, xin[:, :, :2], torch.tensor([980]), cond[:, :2], win=40)
        for nt in (16, 32, 64, 128):
            torch.set_num_threads(nt)
            t0 = time.time(); O.unet_forward(sd, xin, torch.tensor([980]), cond, win=40); dt = time.time() - t0
            res[f"oracle_T{Ts}_h{h}_threads{nt}_s"] = dt
            print(f"oracle T={Ts} h={h} threads={nt}: {dt:.2f} s", flush=True)
print(json.dumps(res))
