"""T-sharding of one long clip across the GPUs of a node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed inference; this is the build's own parallelism (§8e):
contiguous frame ranges per rank, and exactly the three exchanges parity requires --
  1. temporal-attention halo: each of the 10 temporal attentions needs `win` neighbour frames of its
     (pre-LayerNorm) input on each side -> point-to-point send/recv with the T-neighbours only
     (xGMI is point-to-point: one link per neighbour pair, no ring/all-to-all traffic);
  2. GroupNorm(8) statistics span all frames (MT:230,235) -> 128-byte fp64 all-reduce per GroupNorm;
  3. the dynamic-threshold quantile is over the whole clip (MT:1186-1190) -> all-reduce of the radix-select
     histograms (3 x <= 8 KB of counters) + one 4-byte MIN.
Everything else of the path is frame-local.  Noise is drawn from a counter-based generator keyed by the
global element index, so results do not depend on the sharding.
"""
from __future__ import annotations

from typing import Tuple

import torch

Tensor = torch.Tensor


class TShardComm:
    def __init__(self, dist, rank: int, world: int, Ttotal: int, f0: int, F: int, group=None):
        self.dist, self.rank, self.world = dist, rank, world
        self.Ttotal, self.f0, self.F = Ttotal, f0, F
        self.group = group

    # ---- tiny reductions
    def all_reduce_sum(self, t: Tensor) -> None:
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_reduce_min(self, t: Tensor) -> None:
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)

    def all_gather_cat(self, v: Tensor) -> Tensor:
        """Used only by the torch reference op set in tests (equal shard sizes)."""
        parts = [torch.empty_like(v) for _ in range(self.world)]
        self.dist.all_gather(parts, v.contiguous(), group=self.group)
        return torch.cat(parts)

    # ---- neighbour halo exchange for the windowed temporal attention
    def halo_exchange(self, x: Tensor, HW: int, win: int) -> Tuple[Tensor, int]:
        """x (F*HW, C) own frames -> (xe ((hl+F+hh)*HW, C), q0=hl) with hl/hh = win frames from the lower /
        upper T-neighbour (0 at the clip ends)."""
        F = x.shape[0] // HW
        if F < win and self.world > 1:
            raise ValueError(f"T-shard needs at least win={win} frames per rank, got {F}")
        lo, hi = self.rank > 0, self.rank < self.world - 1
        hl, hh = (win if lo else 0), (win if hi else 0)
        C = x.shape[1]
        xe = torch.empty((hl + F + hh) * HW, C, device=x.device, dtype=x.dtype)
        xe[hl * HW:(hl + F) * HW].copy_(x)
        ops = []
        d = self.dist
        if lo:
            ops.append(d.P2POp(d.isend, x[:win * HW], self.rank - 1, group=self.group))
            ops.append(d.P2POp(d.irecv, xe[:hl * HW], self.rank - 1, group=self.group))
        if hi:
            ops.append(d.P2POp(d.isend, x[(F - win) * HW:], self.rank + 1, group=self.group))
            ops.append(d.P2POp(d.irecv, xe[(hl + F) * HW:], self.rank + 1, group=self.group))
        if ops:
            for w in d.batch_isend_irecv(ops):
                w.wait()
        return xe, hl
