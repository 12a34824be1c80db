"""Test infrastructure for `-m gpu` T-shard tests on ONE GPU: several ranks of a T-sharded clip run in host threads of this process
(one HIP stream each) and exchange through a barrier what RCCL would move over xGMI -- no RCCL, no gloo.  Two front ends:

  * `Exchange.callbacks(rank)`   -> ctx.ShardCallbacks for the C-ABI rank (dawn_unet_forward_sharded / dawn_sampler_run_sharded);
  * `InProcComm(ex, rank, world, F)` -> a tshard.TShardComm for the Python-orchestrated rank (unet_forward / sampler), same schedule
    as over RCCL (edge-first producers, halo_begin / halo_end, GroupNorm and histogram all-reduces), the transfer done by device copies.
"""
import threading

import torch

from dawn_pytorch_amd.ctx import ShardCallbacks
from dawn_pytorch_amd.tshard import TShardComm


def _sync():
    torch.cuda.current_stream().synchronize()


class Exchange:
    """What RCCL would do, between `world` host threads on one GPU (equal contiguous shards, one neighbour per halo: F >= win)."""

    def __init__(self, world, timeout=120):
        self.world, self.timeout = world, timeout
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def halo(self, rank, v, hl, F, hh):
        """v = [hl | F | hh] frames (2-D view, one row per frame) with this rank's own frames in place: fill the halo rows."""
        _sync()                                                # this rank's own (edge) frames are complete
        self.slots[rank] = (v, hl, F)
        self.barrier.wait(timeout=self.timeout)
        if hl:
            src, shl, sF = self.slots[rank - 1]
            v[:hl].copy_(src[shl + sF - hl:shl + sF])
        if hh:
            src, shl, sF = self.slots[rank + 1]
            v[hl + F:].copy_(src[shl:shl + hh])
        _sync()
        self.barrier.wait(timeout=self.timeout)                # nobody moves on (and overwrites its rows) before the copies are done

    def reduce(self, rank, t, op):
        _sync()
        self.slots[rank] = t.clone()
        self.barrier.wait(timeout=self.timeout)
        tot = self.slots[0].clone()
        for r in range(1, self.world):
            tot = op(tot, self.slots[r])
        self.barrier.wait(timeout=self.timeout)
        t.copy_(tot)
        _sync()

    def callbacks(self, rank):
        ex = self

        def halo_begin(xe, hl, F, hh, frame_floats):
            ex.halo(rank, xe.view(hl + F + hh, frame_floats), hl, F, hh)

        def red(op):
            return lambda t: ex.reduce(rank, t, op)
        return ShardCallbacks(rank, self.world, halo_begin, lambda: None, red(torch.add), red(torch.add), red(torch.minimum))


class InProcComm(TShardComm):
    """tshard.TShardComm whose transport is `Exchange` (the Python host's rank in a thread of this process)."""

    def __init__(self, ex: Exchange, rank: int, F: int):
        super().__init__(None, rank, ex.world, ex.world * F, rank * F, F)
        self.ex = ex

    def all_reduce_sum(self, t):
        self.n_allreduce += 1
        self.allreduce_bytes += t.numel() * t.element_size()
        self.ex.reduce(self.rank, t, torch.add)

    def all_reduce_min(self, t):
        self.n_allreduce += 1
        self.allreduce_bytes += t.numel() * t.element_size()
        self.ex.reduce(self.rank, t, torch.minimum)

    def halo_post(self, xe, hl, F, hh, frame_floats):
        if F != self.F or max(hl, hh) > F:
            raise ValueError("InProcComm: one neighbour per halo (F >= win)")
        self.ex.halo(self.rank, xe.reshape(hl + F + hh, frame_floats), hl, F, hh)
        self.halo_bytes_recv += (hl + hh) * frame_floats * 4
        self.halo_bytes_sent += ((min(self._win, F) if self.rank > 0 else 0) + (min(self._win, F) if self.rank < self.world - 1 else 0)) * frame_floats * 4
        self.n_halo += 1
        return []


def run_ranks(world, fn, timeout=300):
    """fn(rank) in `world` host threads, each on a HIP stream of its own; returns the per-rank results (re-raises a rank's exception)."""
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                out[r] = fn(r)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:                              # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=timeout)
    for e in err:
        if e is not None:
            raise e
    assert all(o is not None for o in out), "a rank thread did not finish"
    return out
