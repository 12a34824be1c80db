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

from typing import Optional, Tuple

import torch

Tensor = torch.Tensor


class HaloExchange:
    """One temporal-attention halo exchange in flight: `xe` = [lower halo | own frames | upper halo] rows, `hl` / `hh` the
    halo frame counts actually present (0 at the clip ends), `works` the outstanding P2P requests."""

    def __init__(self, xe: Tensor, hl: int, hh: int, F: int, works):
        self.xe, self.hl, self.hh, self.F, self.works = xe, hl, hh, F, works

    @property
    def Fext(self) -> int:
        return self.hl + self.F + self.hh


class TShardComm:
    """Contiguous, equal frame ranges per rank: rank r owns the global frames [r*F, (r+1)*F) of a clip of Ttotal = world*F.

    The halo exchange is SPLIT in two calls so that the orchestration can compute while the transfer runs
    (`halo_begin` posts the sends / receives and returns, `halo_end` makes the current stream wait for them): everything
    that only needs this rank's own frames -- the queries whose window does not reach a neighbour, the qkv projection of
    the own rows -- is launched in between (unet_forward._temporal).  A halo wider than the neighbour's shard (F < win) is
    gathered from as many ranks as it spans.  Buffers are cached per shape (no allocation per layer and step)."""

    def __init__(self, dist, rank: int, world: int, Ttotal: int, f0: int, F: int, group=None, reduce_group=None):
        """group = the process group of the halo point-to-point exchanges (None: the default group); reduce_group = the group of the
        tiny GroupNorm / quantile all-reduces (None: the same as `group`).  With torch's NCCL (= RCCL) backend every collective of
        one process group runs on that group's communicator stream: a 128-byte GroupNorm all-reduce issued while a 186 MB halo
        transfer is in flight would queue BEHIND it -- exactly what the edge-first schedule (unet_forward._edge_first) tries to
        overlap.  Two groups = two communicators = two streams: `TShardComm.two_groups(dist)` creates them."""
        self.dist, self.rank, self.world = dist, rank, world
        self.Ttotal, self.f0, self.F = Ttotal, f0, F
        self.group = group
        self.reduce_group = group if reduce_group is None else reduce_group
        self.timing = False           # measure what the stream waits for (HIP events around halo_end / the all-reduces): bench.py's
        self._timed = []              # ... extra clip after the timed region; [(kind, start event, end event)] or host seconds on CPU
        self._host_s = {"halo_wait": 0.0, "allreduce": 0.0}
        if world > 1 and (f0 != rank * F or Ttotal != world * F):
            raise ValueError("TShardComm expects equal contiguous shards: f0 == rank*F and Ttotal == world*F")
        self._bufs = {}
        # long shards (the memory-lean form of an evaluation) do not keep the extended buffers between layers: cached per shape they pin
        # 2.4 MB per frame at 256 x 256 (one (Fext*HW, C) buffer per level and width) -- half as much again as the evaluation itself
        # needs (measured: 7.2 instead of 4.8 MB per own frame); short shards keep them (no allocation per layer and step)
        self.keep_buffers = True
        self._win = 0                 # attention window of the exchanges (set by halo_begin / set_window)
        self.n_halo = self.n_allreduce = 0
        self.n_halo_edge_first = 0    # exchanges posted by the PRODUCER of the layer input (unet_forward._edge_first), own rows in place
        self.halo_bytes_sent = self.halo_bytes_recv = self.allreduce_bytes = 0

    @staticmethod
    def two_groups(dist):
        """(halo group, reduce group): two process groups over all ranks, so that the point-to-point halo transfers and the tiny
        all-reduces get a communicator (and, on RCCL, a stream) each.  Collective: every rank calls it, in the same order."""
        ranks = list(range(dist.get_world_size()))
        return dist.new_group(ranks), dist.new_group(ranks)

    def stats(self) -> dict:
        """Counters since construction (bench.py reports them per rank: RCCL participation is checkable from the JSON)."""
        return {"rank": self.rank, "world": self.world, "halo_exchanges": self.n_halo, "halo_exchanges_edge_first": self.n_halo_edge_first,
                "halo_bytes_sent": self.halo_bytes_sent, "halo_bytes_received": self.halo_bytes_recv,
                "all_reduces": self.n_allreduce, "all_reduce_bytes": self.allreduce_bytes,
                "separate_groups": self.reduce_group is not self.group}

    # ---- what the compute stream waits for (timing = True only; bench.py runs ONE extra clip with it after the timed region)
    def _timed_call(self, kind: str, like: Optional[Tensor], fn) -> None:
        if not self.timing:
            fn()
            return
        if like is not None and like.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            self._timed.append((kind, e0, e1))
        else:
            import time
            t0 = time.perf_counter()
            fn()
            self._host_s[kind] += time.perf_counter() - t0

    def timing_ms(self, reset: bool = True) -> dict:
        """{"halo_wait_ms", "allreduce_ms", ...}: time the compute stream spent between the HIP events recorded around every
        halo_end (= how long it WAITED for the neighbours' frames: 0 when the transfer hid behind the producer) and around every
        all-reduce (launch + latency of the collective on the critical path) since `timing` was switched on.  Synchronises."""
        out = {"halo_wait_ms": self._host_s["halo_wait"] * 1e3, "allreduce_ms": self._host_s["allreduce"] * 1e3,
               "halo_waits": 0, "allreduces_timed": 0}
        if self._timed:
            torch.cuda.synchronize()
            for kind, e0, e1 in self._timed:
                out["halo_wait_ms" if kind == "halo_wait" else "allreduce_ms"] += e0.elapsed_time(e1)
                out["halo_waits" if kind == "halo_wait" else "allreduces_timed"] += 1
        if reset:
            self._timed = []
            self._host_s = {"halo_wait": 0.0, "allreduce": 0.0}
        return out

    # ---- tiny reductions
    def all_reduce_sum(self, t: Tensor) -> None:
        self.n_allreduce += 1
        self.allreduce_bytes += t.numel() * t.element_size()
        self._timed_call("allreduce", t, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.reduce_group))

    def all_reduce_min(self, t: Tensor) -> None:
        self.n_allreduce += 1
        self.allreduce_bytes += t.numel() * t.element_size()
        self._timed_call("allreduce", t, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.reduce_group))

    def all_gather_cat(self, v: Tensor) -> Tensor:
        """Used only by the torch reference op set in tests (equal shard sizes)."""
        parts = [torch.empty_like(v) for _ in range(self.world)]
        self.dist.all_gather(parts, v.contiguous(), group=self.group)
        return torch.cat(parts)

    # ---- neighbour halo exchange for the windowed temporal attention
    def _buffer(self, rows: int, C: int, like: Tensor) -> Tensor:
        key = (rows, C, like.device, like.dtype)
        b = self._bufs.get(key)
        if b is None:
            b = self._bufs[key] = torch.empty(rows, C, device=like.device, dtype=like.dtype)
        return b

    def own_buffer(self, F: int, HW: int, C: int, win: int, like: Tensor) -> Tensor:
        """The cached extended buffer [lower halo | own | upper halo] for an (F*HW, C) layer input."""
        lo_g, hi_g = max(0, self.f0 - win), min(self.Ttotal, self.f0 + F + win)
        return self._buffer((hi_g - lo_g) * HW, C, like)

    def own_view(self, F: int, HW: int, C: int, win: int, like: Tensor) -> Tensor:
        """The own-rows slice of that buffer: a PRODUCER that writes the temporal layer's input straight into it (out=...)
        saves halo_begin's copy of the own rows (210 MB per level-0 layer at 256 x 256).  Valid until the next exchange of the
        same shape (the buffer is cached per shape), i.e. for a tensor whose only consumer is that temporal layer."""
        hl = self.f0 - max(0, self.f0 - win)
        return self.own_buffer(F, HW, C, win, like)[hl * HW:(hl + F) * HW]

    def halo_begin(self, x: Tensor, HW: int, win: int) -> HaloExchange:
        """x (F*HW, C) own frames.  Copies them into the middle of the (cached) extended buffer (unless a producer already wrote them
        there: own_view) and posts the point-to-point sends / receives of the halo frames: the lower halo = global frames
        [f0 - win, f0), the upper = [f0 + F, f0 + F + win), clipped to the clip, each piece from the rank that owns it.
        NOTE the returned buffer is CACHED per (rows, C): it stays valid only until the next exchange of the same shape on this
        communicator (the three 64-channel level-0 layers share one); callers consume it before the next exchange (stream
        order).  `release_buffers()` drops the cache between clips."""
        F = x.shape[0] // HW
        if F != self.F or x.shape[0] != F * HW:
            raise ValueError(f"halo_begin: {x.shape[0]} rows of {HW} pixels are not this rank's {self.F} frames")
        C = x.shape[1]
        lo_g, hi_g = max(0, self.f0 - win), min(self.Ttotal, self.f0 + F + win)     # global frame range of the buffer
        hl, hh = self.f0 - lo_g, hi_g - (self.f0 + F)
        self._win = win
        xe = self._buffer((hl + F + hh) * HW, C, x)
        if x.data_ptr() != xe[hl * HW:].data_ptr():          # (already in place when the producer wrote into own_view())
            xe[hl * HW:(hl + F) * HW].copy_(x)
        return HaloExchange(xe, hl, hh, F, self.halo_post(xe, hl, F, hh, HW * C))

    def halo_post(self, xe: Tensor, hl: int, F: int, hh: int, frame_floats: int):
        """Post the sends / receives for an extended buffer whose own frames are in place: xe = [hl | F | hh] frames of
        frame_floats floats (any 2-D / 1-D contiguous float view).  Returns the outstanding requests (halo_end waits for them).
        Also the body of the C evaluator's halo_begin callback (ctx.ShardCallbacks.from_tshard), after set_window(win)."""
        if F != self.F:
            raise ValueError(f"halo_post: {F} own frames, but this rank holds {self.F}")
        if self._win < max(hl, hh):
            raise ValueError("halo_post: set_window(win) first (the window tells what the neighbours need from this rank)")
        if xe.numel() != (hl + F + hh) * frame_floats:
            raise ValueError(f"halo_post: buffer of {xe.numel()} floats is not [{hl} | {F} | {hh}] frames of {frame_floats}")
        xe = xe.reshape(hl + F + hh, frame_floats)
        lo_g, hi_g = self.f0 - hl, self.f0 + F + hh
        d = self.dist
        ops = []
        esz = xe.element_size() * frame_floats
        for r in range(self.world):
            if r == self.rank:
                continue
            r0, r1 = r * F, (r + 1) * F                                             # frames rank r owns
            # what I receive from r: its frames inside my buffer range
            a, b = max(r0, lo_g), min(r1, hi_g)
            if a < b:
                ops.append(d.P2POp(d.irecv, xe[a - lo_g:b - lo_g], r, group=self.group))
                self.halo_bytes_recv += (b - a) * esz
            # what I send to r: my frames inside ITS buffer range [r0 - win_r, r1 + win_r), win_r = its halo widths
            wl, wh = min(self._win, r0), min(self._win, self.Ttotal - r1)
            a, b = max(self.f0, r0 - wl), min(self.f0 + F, r1 + wh)
            if a < b:
                ops.append(d.P2POp(d.isend, xe[a - lo_g:b - lo_g], r, group=self.group))
                self.halo_bytes_sent += (b - a) * esz
        works = d.batch_isend_irecv(ops) if ops else []
        self.n_halo += 1
        return works

    def set_window(self, win: int) -> None:
        """Attention window of the exchanges (what the NEIGHBOURS need from this rank; halo_begin sets it itself)."""
        self._win = win

    def release_buffers(self) -> None:
        """Drop the cached extended buffers (they pin (Fext*HW, C) per level for the lifetime of the communicator)."""
        self._bufs.clear()

    def wait_works(self, works, like: Optional[Tensor] = None) -> None:
        """NCCL/RCCL: the current stream waits for the transfers; gloo: the host does (also the body of the C evaluator's halo_end)."""
        def go():
            for w in works:
                w.wait()
        self._timed_call("halo_wait", like, go)

    def halo_end(self, hx: HaloExchange) -> None:
        self.wait_works(hx.works, hx.xe)
        hx.works = []
        if not self.keep_buffers:
            self._bufs.clear()        # (hx.xe keeps the buffer alive until the layer that reads it has been enqueued)

    def halo_exchange(self, x: Tensor, HW: int, win: int) -> Tuple[Tensor, int]:
        """Blocking form: (xe, first own frame index in xe).  xe is a private copy (safe to keep), unlike halo_begin's."""
        hx = self.halo_begin(x, HW, win)
        self.halo_end(hx)
        return hx.xe.clone(), hx.hl


class SimulatedInteriorShard(TShardComm):
    """ONE interior rank's workload of a T-sharded clip, on one GPU (bench.py `shard_sim`): the rank owns frames
    [f0, f0 + F) of a clip of Ttotal = world * F frames with a neighbour on each side, so every temporal attention runs on
    [lower halo | own | upper halo] rows, the GroupNorm statistics take the separate reduce -> all-reduce -> finalize path and the
    threshold selection the histogram all-reduces.  The halo rows are FILLED LOCALLY (copies of this rank's own edge frames: the
    values are irrelevant to the cost, the bytes moved on the device are not the link's) and the all-reduces run on a
    world-size-1 communicator when `dist` is given (their launch + latency is real, their payload trivial) or are skipped.
    What it measures: the compute-side price of the sharded shape -- segmentation, halo-row projections, own-row placement,
    extra small kernels -- against the unsharded clip of the same F frames.  Link time is NOT in it."""

    def __init__(self, F: int, world: int = 8, rank: int = 3, dist=None):
        super().__init__(dist, rank, world, world * F, rank * F, F)
        self.simulated = True

    def all_reduce_sum(self, t: Tensor) -> None:
        self.n_allreduce += 1
        self.allreduce_bytes += t.numel() * t.element_size()
        if self.dist is not None:
            self._timed_call("allreduce", t, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM))

    def all_reduce_min(self, t: Tensor) -> None:
        self.n_allreduce += 1
        self.allreduce_bytes += t.numel() * t.element_size()
        if self.dist is not None:
            self._timed_call("allreduce", t, lambda: self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN))

    def halo_begin(self, x: Tensor, HW: int, win: int) -> HaloExchange:
        F = x.shape[0] // HW
        C = x.shape[1]
        lo_g, hi_g = max(0, self.f0 - win), min(self.Ttotal, self.f0 + F + win)
        hl, hh = self.f0 - lo_g, hi_g - (self.f0 + F)
        xe = self.own_buffer(F, HW, C, win, x)
        if x.data_ptr() != xe[hl * HW:].data_ptr():
            xe[hl * HW:(hl + F) * HW].copy_(x)
        n = min(hl, F)
        if n:
            xe[(hl - n) * HW:hl * HW].copy_(x[(F - n) * HW:])        # "received" lower halo: the own last frames
        n = min(hh, F)
        if n:
            xe[(hl + F) * HW:(hl + F + n) * HW].copy_(x[:n * HW])    # "received" upper halo: the own first frames
        self.halo_bytes_recv += (hl + hh) * HW * C * x.element_size()
        self.halo_bytes_sent += (min(win, F) * 2) * HW * C * x.element_size()
        self.n_halo += 1
        return HaloExchange(xe, hl, hh, F, [])
