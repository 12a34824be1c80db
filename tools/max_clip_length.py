#!/usr/bin/env python3
"""GPU tool: max clip length in HBM on ONE GPU at 256x256 (second half of BASELINE.json's metric).
Measures peak allocator memory of one full DDIM step (UNet evaluation + threshold + update) at a few clip
lengths, fits bytes/frame, then PROVES a long clip by actually running one step at `--try-frames`."""
import argparse, json, os, sys, time
# (no allocator environment variable: the caching allocator's default block splitting fragments the pool on clips this long -- 52,000
#  frames pass, 56,000 fail with 55 GiB reserved-but-unallocated -- so the package itself switches the splitting of blocks > 2 GiB off
#  when it meets a long clip: diffusion._long_clip_allocator, called below as GaussianDiffusion.ddim_sample calls it)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dawn_pytorch_amd.sampler import ddim_sample_clip, ddim_step_scalars
from dawn_pytorch_amd.diffusion import _long_clip_allocator
from dawn_pytorch_amd.tshard import SimulatedInteriorShard

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--probe", default="4800,6400", help="above unet_forward.LONG_CLIP_FRAMES: the memory-lean form of long clips")
ap.add_argument("--try-frames", type=int, default=0, help="0 = 85 %% of the extrapolated limit")
ap.add_argument("--host", choices=["python", "ctx"], default="python",
                help="ctx = the C-side evaluator (dawn_sampler_run): two caller-owned allocations, no caching-allocator fragmentation")
ap.add_argument("--shard-sim", action="store_true",
                help="the clip is ONE INTERIOR RANK's shard of an 8-way T-sharded clip (tshard.SimulatedInteriorShard: own frames + 2 x win halo "
                     "frames at every temporal attention, GroupNorm reduce / finalize split, histogram path): bytes per OWN frame of a rank")
a = ap.parse_args()
dev = torch.device("cuda", 0)
h = a.res // 4
total = torch.cuda.get_device_properties(0).total_memory


def one_step(T):
    unet, diff = bench.build_model(T, h, 50, dev)
    diff.noise_seed = 1
    fea, bbox, cond = bench.synthetic_inputs(T, h, dev)
    _long_clip_allocator(T, fea)
    ops = unet._ops()
    P = unet.packed()
    comm = SimulatedInteriorShard(T, world=8, rank=3) if a.shard_sim else None
    if comm is not None:
        assert a.host == "python", "--shard-sim runs the Python-orchestrated rank"
        ops = ops.with_comm(comm)
    f0, Ttot = (comm.f0, comm.Ttotal) if comm is not None else (0, T)
    cs = None if a.host == "ctx" else unet.build_clip(torch.cat((fea, bbox), 1)[0].contiguous(), cond[0].contiguous(), comm=comm,
                                                       Ttotal=Ttot, f0=f0)
    steps = ddim_step_scalars({k: getattr(diff, k) for k in ("alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                                                              "sqrt_recipm1_alphas_cumprod")}, 50, 1.0)[:1]
    x0 = ops.philox_normal(3, T, f0, Ttot, h * h, 1, 0, dev).reshape(3, T, h, h)
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    t0 = time.time()
    if a.host == "ctx":
        ev = unet.ctx_evaluator()
        fea272 = torch.cat((fea, bbox), 1)[0].contiguous()
        clip = ev.prepare_clip(fea272, cond[0].contiguous())
        out = ev.sample(clip, x0, steps, seed=1)
        del clip, ev
    else:
        out = ddim_sample_clip(ops, P, cs, x0, steps, lambda i: ops.philox_normal(3, T, f0, Ttot, h * h, 1, i + 1, dev).reshape(3, T, h, h))
        if comm is not None:
            comm.release_buffers()
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert torch.isfinite(out).all()
    peak = torch.cuda.max_memory_allocated()
    del unet, diff, cs, out, x0, comm, ops
    torch.cuda.empty_cache()
    return peak, dt


res = {"resolution": a.res, "hbm_bytes": total, "probes": [], "mode": "one interior rank of an 8-way T-shard (halos filled locally)" if a.shard_sim else "unsharded clip",
       "allocator": "set by the package at run time (diffusion._long_clip_allocator), no environment variable"}
pts = []
for T in [int(t) for t in a.probe.split(",")]:
    peak, dt = one_step(T)
    pts.append((T, peak))
    res["probes"].append({"frames": T, "peak_bytes": peak, "step_seconds": dt})
    print(f"T={T}: peak {peak/2**30:.2f} GiB, one DDIM step {dt:.2f} s", flush=True)
(T1, p1), (T2, p2) = pts[-2], pts[-1]
per_frame = (p2 - p1) / (T2 - T1)
fixed = p2 - per_frame * T2
limit = int((total * 0.97 - fixed) / per_frame)
res.update(bytes_per_frame=per_frame, fixed_bytes=fixed, extrapolated_max_frames=limit)
Ttry = a.try_frames or int(limit * 0.85)
try:
    peak, dt = one_step(Ttry)
    res["proved"] = {"frames": Ttry, "peak_bytes": peak, "step_seconds": dt}
    print(f"PROVED T={Ttry}: peak {peak/2**30:.1f} GiB of {total/2**30:.1f} GiB, one DDIM step {dt:.1f} s")
except Exception as e:                      # noqa: BLE001
    res["proved"] = {"frames": Ttry, "error": str(e)[:200]}
    print("FAILED at", Ttry, e)
print(json.dumps(res))
