#!/usr/bin/env python3
"""GPU smoke of the T-shard path over RCCL: launch with torchrun on a box with at least WORLD_SIZE GPUs (RCCL refuses two ranks on one
device: "Duplicate GPU detected", tried on the 1-GPU boxes of this build's pool in round 4 -- the multi-rank path is covered there by the
in-process ranks of tests/test_hip_shard_fullsize.py and by the gloo tests on CPU).
Checks that the sharded 2-rank result equals the single-rank result of the same clip (tiny model)."""
import os, sys, datetime
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dawn_pytorch_amd as D
from dawn_pytorch_amd.tshard import TShardComm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ndev = torch.cuda.device_count()
if ndev < world:
    sys.exit(f"tshard_gpu_smoke: {world} ranks need {world} GPUs, {ndev} visible (RCCL refuses two ranks on one device)")
dev = torch.device("cuda", rank % ndev)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=120), device_id=dev)
KW = dict(dim=64, cond_dim=40, cond_aud=32, cond_pose=6, cond_eye=2, channels=35, dim_mults=(1, 2), use_hubert_audio_cond=True, win_width=8)
TT, h, S = 48, 16, 3
F = TT // world
def build(T):
    unet = D.DynamicNfUnet3D(default_num_frames=T, num_frames=T, **KW).to(dev)
    diff = D.DynamicNfGaussianDiffusion(default_num_frames=T, denoise_fn=unet, num_frames=T, image_size=h, sampling_timesteps=S, use_dynamic_thres=True).to(dev)
    diff.noise_seed = 5
    return diff
g = torch.Generator().manual_seed(3)
fea, bbox, cond = torch.randn(1, 28, h, h, generator=g).to(dev), torch.randn(1, 4, h, h, generator=g).to(dev), torch.randn(1, TT, 40, generator=g).to(dev)
comm = TShardComm(dist, rank, world, TT, rank * F, F)
out = build(F).sample(fea, bbox, cond=cond[:, rank * F:(rank + 1) * F].contiguous(), comm=comm)
parts = [torch.empty_like(out) for _ in range(world)]
dist.all_gather(parts, out)
if rank == 0:
    full = build(TT).sample(fea, bbox, cond=cond)
    err = float((torch.cat(parts, 2) - full).abs().max())
    print(f"TSHARD_GPU_SMOKE world={world} max|sharded - single| = {err:.3e}")
    assert err < 1e-4
dist.barrier()
dist.destroy_process_group()
