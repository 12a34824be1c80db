#!/usr/bin/env python3
"""GPU: where does the allocator peak of one UNet evaluation on the long-clip path occur?  Wraps HipOps.empty, records the allocated
bytes and the call stack at every allocation, prints the top moments."""
import os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dawn_pytorch_amd.unet_forward import unet_forward
dev = torch.device("cuda", 0)
T, h = int(sys.argv[1]) if len(sys.argv) > 1 else 320, 64
SHARD = len(sys.argv) > 2 and sys.argv[2] == "shard"          # one interior rank of an 8-way T-shard (halos filled locally)
unet, diff = bench.build_model(200, h, 50, dev)
ops, P = unet._ops(), unet.packed()
fea, bbox, cond = bench.synthetic_inputs(T, h, dev)
comm = None
if SHARD:
    from dawn_pytorch_amd.tshard import SimulatedInteriorShard
    comm = SimulatedInteriorShard(T, world=8, rank=3)
    ops = ops.with_comm(comm)
    _o = torch.empty
    def _te(*a, **k):                                         # the communicator allocates its extended buffers with torch.empty
        t = _o(*a, **k)
        if t.is_cuda and t.numel() > 1 << 22 and log is not None:
            st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-6:-1]]
            log.append((torch.cuda.memory_allocated() - base, t.numel() * t.element_size(), "torch.empty < " + " < ".join(reversed(st))))
        return t
cs = unet.build_clip(torch.cat((fea, bbox), 1)[0].contiguous(), cond[0].contiguous(), comm=comm, Ttotal=comm.Ttotal if comm else None,
                     f0=comm.f0 if comm else 0)
log = None
x0 = torch.randn(3, T, h, h, device=dev)
unet_forward(ops, P, cs, x0, 500)
torch.cuda.synchronize()
base = torch.cuda.memory_allocated()
log = []
orig = type(ops).empty
def empty(self, *a, **k):
    t = orig(self, *a, **k)
    st = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-6:-1]]
    log.append((torch.cuda.memory_allocated() - base, t.numel() * 4, " < ".join(reversed(st))))
    return t
type(ops).empty = empty
if SHARD:
    torch.empty = _te
unet_forward(ops, P, cs, x0, 500)
torch.cuda.synchronize()
frame_mb = T * h * h * 64 * 4 / 1e6
print(f"T={T}: one level-0 tensor = {frame_mb:.0f} MB; baseline (weights, clip tables, x) {base / 1e6:.0f} MB")
for i, (tot, sz, st) in sorted(enumerate(log), key=lambda kv: -kv[1][0])[:8]:
    print(f"alloc #{i}: live {tot / 1e6:8.0f} MB = {tot / 1e6 / frame_mb:5.2f} level-0 tensors  (+{sz / 1e6:.0f} MB)  {st}")
