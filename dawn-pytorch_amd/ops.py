"""`HipOps`: the op set of the denoising path, each method one (or a few) launches into libdawn_hip.so.

Tensors are torch CUDA(=HIP) fp32 tensors used only as typed device buffers (allocation + stream come from
PyTorch; no torch arithmetic happens here).  Activations are 2-D `(rows, C)` views of channels-last clips,
rows = F*H*W, unit stride along C, arbitrary row stride (so channel slices of wider buffers are legal).

The orchestration in :mod:`unet_forward` / :mod:`sampler` is written against this interface; tests inject
a torch reference implementation with the same interface (oracle/ops_ref.py) to check the orchestration
on CPU and to check every kernel individually on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, check

Tensor = torch.Tensor


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ld(t: Optional[Tensor]) -> int:
    return 0 if t is None else (t.stride(0) if t.dim() >= 2 else t.numel())


def _need(ok, what: str) -> None:
    """Launch precondition (contiguity, shapes, the dtype a pointer offset is computed in): raised, never `assert`ed -- under
    `python -O` an assert disappears and the raw pointers would reach the kernel unchecked."""
    if not ok:
        raise _lib.DawnHipError(f"HipOps precondition failed -- {what}")


class HipOps:
    """Launches on the current PyTorch stream of the tensors' device."""

    name = "hip"

    def __init__(self, comm=None):
        self.L = _lib.lib()          # raises if the extension is missing: no fallback by design
        self.comm = comm             # T-shard communicator (see tshard.py) or None
        self.prof = None             # list -> (algorithmic flops, start event, end event) per conv_gemm launch
        self.prof_layers = None      # list -> (op, shape, start event, end event) per fused-layer launch (bench.py)
        self.prof_on = True          # sampling switch (the sampler records events on every n-th DDIM step only)
        self.overlap = True          # two-stream overlap of independent branches (fork_join)
        self._side_stream = None
        self.graph_error = None      # set when a HIP-graph capture failed and the sampler fell back to eager
        self.conv_policy = 0         # dawn_conv_desc.policy of every conv_gemm launch (0 = shipped kernel policy)
        self.temporal_flags = 0      # kernel-family selector of the fused temporal layer (0 = automatic; A/B and tests)
        self.temporal_attn_flags = 0  # dawn_temporal_attn_ex flags (1 = the fp32-MFMA attention core; A/B and tests)
        self.sk_ws = None            # experimental library only (tools/build_sk_timing_lib.sh): scratch tensor handed to dawn_conv_desc.sk_ws
        self.fuse_h1 = True          # cross-attention kernels write h1 = SiLU(GN(c1)) + h_cond themselves (False: A/B, two-stream form)
        self._sel_ws = {}            # (device index, stream) -> scratch of the threshold selection (histograms, state)
        self._tickets = {}           # (device index, stream) -> the zeroed device word of the convs' fused GroupNorm finalisation
        self.fuse_gn = True          # GroupNorm coefficients from the conv launch itself where the kernel can (False: A/B, separate launch)
        # (keyed by stream as well: the dicts are shared by every with_comm() copy, and two samplers on one device -- in-process
        # ranks, concurrent clips -- run on different streams and must not share histograms / hand-off tickets)

    def _gn_ticket(self, like: Tensor) -> Tensor:
        key = (like.device.index, self._stream())
        t = self._tickets.get(key)
        if t is None:
            t = self._tickets[key] = torch.empty(4, device=like.device, dtype=torch.int32)
            check(self.L.dawn_gn_ticket_reset(_p(t), self._stream()), "dawn_gn_ticket_reset")
        return t

    def begin_evaluation(self, like: Tensor) -> None:
        """Start of one denoiser evaluation on the current stream: the hand-off word of the convs' fused GroupNorm finalisation is
        zeroed by a stream-ordered fill (include/dawn_hip.h: hard precondition of dawn_conv_desc.gn_ticket).  Inside a HIP-graph
        capture this records a memset node, so every replay starts from a zero ticket too; an aborted launch or capture can no
        longer leave a stale count behind for the next evaluation (the C evaluator does the same, dawn_ctx.hip)."""
        check(self.L.dawn_gn_ticket_reset(_p(self._gn_ticket(like)), self._stream()), "dawn_gn_ticket_reset")

    def with_comm(self, comm):
        o = HipOps(comm)
        o.prof = self.prof
        o.prof_layers = self.prof_layers
        o.prof_on = self.prof_on
        o.prof_every = getattr(self, "prof_every", 1)
        o.overlap = self.overlap
        o.conv_policy = self.conv_policy
        o.temporal_flags = self.temporal_flags
        o.temporal_attn_flags = self.temporal_attn_flags
        o.sk_ws = self.sk_ws
        o.fuse_h1 = self.fuse_h1
        o._sel_ws = self._sel_ws
        o._tickets = self._tickets
        o.fuse_gn = self.fuse_gn
        return o

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _require(*ts: Optional[Tensor]) -> None:
        for t in ts:
            if t is None:
                continue
            if not t.is_cuda:
                raise _lib.DawnHipError("HipOps needs GPU tensors; there is no CPU fallback on the product path")
            if t.dtype not in (torch.float32, torch.float64, torch.int32, torch.uint8) or t.stride(-1) != 1:
                raise _lib.DawnHipError(f"bad tensor for HipOps: dtype={t.dtype} strides={t.stride()}")

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def empty(self, *shape, like: Tensor, dtype=torch.float32) -> Tensor:
        return torch.empty(*shape, device=like.device, dtype=dtype)

    def fork_join(self, side, main):
        """Run `side()` on a second HIP stream concurrently with `main()` on the current stream; both see all
        work enqueued so far, and everything enqueued afterwards sees both (stream/event plumbing only)."""
        if not self.overlap:
            a = side()
            return a, main()
        cur = torch.cuda.current_stream()
        if self._side_stream is None or self._side_stream.device != cur.device:
            self._side_stream = torch.cuda.Stream(device=cur.device)
        s2 = self._side_stream
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            a = side()
        b = main()
        cur.wait_stream(s2)
        for t in (a if isinstance(a, (tuple, list)) else (a,)):
            if torch.is_tensor(t):
                t.record_stream(cur)       # allocated on the side stream, consumed on the main one
        return a, b

    # ------------------------------------------------------------------ conv / linear on MFMA
    def conv_gemm(self, in0: Tensor, w: Tensor, N: int, *, F: int, Hi: int, Wi: int, Ho: Optional[int] = None,
                  Wo: Optional[int] = None, KH: int = 1, KW: int = 1, stride: int = 1, pad: int = 0, mode: int = 0,
                  in1: Optional[Tensor] = None, bias: Optional[Tensor] = None,
                  row_stats: Optional[Tuple[Tensor, Tensor]] = None, ch_ab: Optional[Tuple[Tensor, Tensor]] = None,
                  pro_act: int = 0, pro_add: Optional[Tensor] = None, res: Optional[Tensor] = None,
                  tr: Optional[Tuple[Tensor, Tensor, Tensor]] = None, out: Optional[Tensor] = None,
                  gn_part: Optional[Tensor] = None, w_bf3: Optional[Tensor] = None, ln_eps: float = 0.0,
                  w_wino: Optional[Tensor] = None, gn_fin: Optional[tuple] = None, w_wino4: Optional[Tensor] = None) -> Tensor:
        """gn_fin = (gamma, beta, film or None, total_rows[, eps]) with gn_part: ask the launch to finish the GroupNorm itself (the
        Winograd 3x3 kernel's last workgroup reduces and finalises); gn_coeffs(part=...) then returns its coefficients without a launch."""
        Ho = Hi if Ho is None else Ho
        Wo = Wi if Wo is None else Wo
        rows_out = F * Ho * Wo
        if out is None:
            out = self.empty(rows_out, N, like=in0)
        self._require(in0, in1, w, bias, pro_add, res, out)
        d = ConvDesc()
        d.in0, d.in1 = _p(in0), _p(in1)
        d.C0, d.C1 = in0.shape[1], (in1.shape[1] if in1 is not None else 0)
        d.ld0, d.ld1 = _ld(in0), _ld(in1)
        d.F, d.Hi, d.Wi, d.Ho, d.Wo = F, Hi, Wi, Ho, Wo
        d.KH, d.KW, d.stride, d.pad, d.mode = KH, KW, stride, pad, mode
        d.w, d.bias, d.N = _p(w), _p(bias), N
        if row_stats is not None:
            d.row_mean, d.row_rstd = _p(row_stats[0]), _p(row_stats[1])
        if ch_ab is not None:
            d.ch_a, d.ch_b = _p(ch_ab[0]), _p(ch_ab[1])
        d.pro_act = pro_act
        d.pro_add, d.ld_add = _p(pro_add), _ld(pro_add)
        d.res, d.ld_res = _p(res), _ld(res)
        if tr is not None:
            d.tr, d.ld_tr, d.tr_a, d.tr_b = _p(tr[0]), _ld(tr[0]), _p(tr[1]), _p(tr[2])
        d.out, d.ld_out = _p(out), _ld(out)
        d.gn_part = _p(gn_part)
        d.w_bf3 = _p(w_bf3)
        d.w_wino = _p(w_wino)
        d.w_wino4 = _p(w_wino4)
        d.policy = self.conv_policy
        d.ln_eps = ln_eps
        if self.sk_ws is not None and w_bf3 is not None and KH == 3 and KW == 3 and stride == 1 and mode == 0:
            d.sk_ws, d.sk_ws_bytes = _p(self.sk_ws), self.sk_ws.numel()        # (ignored by the shipped library)
        nrows = C.c_int(0)
        fin_ab = None
        if gn_part is not None:
            d.gn_rows = C.pointer(nrows)
            gn_part.dawn_ab = None
            if gn_fin is not None and self.comm is None and self.fuse_gn:
                gamma, beta, film, total_rows = gn_fin[:4]
                fin_ab = (self.empty(N, like=in0), self.empty(N, like=in0))
                d.gn_gamma, d.gn_beta = _p(gamma), _p(beta)
                if film is not None:
                    d.gn_fs, d.gn_fsh = _p(film[0]), _p(film[1])
                d.gn_count = float(total_rows) * (N // 8)
                d.gn_eps = gn_fin[4] if len(gn_fin) > 4 else 1e-5
                d.gn_a, d.gn_b = _p(fin_ab[0]), _p(fin_ab[1])
                d.gn_ticket = _p(self._gn_ticket(in0))
        if self.prof is not None and self.prof_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(self.L.dawn_conv_gemm(C.byref(d), self._stream()), "dawn_conv_gemm")
            e1.record()
            if gn_part is not None:
                gn_part.dawn_rows = abs(nrows.value)
                gn_part.dawn_ab = fin_ab if nrows.value < 0 else None
            rows_gemm = rows_out if mode == 0 else F * Hi * Wi * 4
            self.prof.append((2.0 * rows_gemm * N * KH * KW * (d.C0 + d.C1), e0, e1,
                              f"M={rows_gemm} N={N} K={KH * KW * (d.C0 + d.C1)} k={KH}x{KW} s={stride} mode={mode} "
                              f"pro={'r' if row_stats else ('n' if ln_eps else '')}{'c' if ch_ab else ''}{'a' if pro_add is not None else ''}"
                              + (" split-bf16" if self._runs_split_kernel(w_bf3, KH, KW, stride, mode, rows_out, N, d.C0, d.C1, tr,
                                                                          gn_part) else "")
                              + ("", " winograd", " winograd4")[self.conv3x3_form(d)],
                              4.0 * (F * Hi * Wi * (d.C0 + d.C1) + rows_out * N + KH * KW * (d.C0 + d.C1) * N * (4 if mode else 1))))
            return out
        check(self.L.dawn_conv_gemm(C.byref(d), self._stream()), "dawn_conv_gemm")
        if gn_part is not None:
            gn_part.dawn_rows = abs(nrows.value)      # rows the launch wrote (the rest of the buffer is unused)
            gn_part.dawn_ab = fin_ab if nrows.value < 0 else None      # ... and the coefficients, when the launch finalised them itself
        return out

    def _runs_split_kernel(self, w_bf3, KH, KW, stride, mode, rows, N, C0, C1, tr, gn_part) -> bool:
        """Profiling label only: does this launch take a split-operand (bf16 pipe) kernel?  3x3/s1 ResBlock convs always do;
        for the 1x1 GEMMs the library's own dispatch predicate is asked (dawn_gemm1x1_split_ok)."""
        if w_bf3 is None:
            return False
        if mode == 1 or (KH == 4 and KW == 4 and stride == 2):      # Downsample / Upsample on the row-accumulator kernel
            return C1 == 0 and C0 % 64 == 0 and N in (64, 128, 256) and gn_part is None and tr is None
        if mode != 0 or stride != 1:
            return False
        if KH == 3 and KW == 3:
            return True
        return KH == 1 and KW == 1 and gn_part is None and self.split_gemm_ok(rows, N, C0, C1)

    def conv3x3_form(self, d) -> int:
        """Which form dawn_conv_gemm runs this descriptor in: 2 = Winograd F(4x4,3x3), 1 = Winograd F(2x2,3x3), 0 = anything else.  The library's
        own launch decision (dawn_conv3x3_form: policy bits, per-shape gates, stride / alignment fallbacks), not a mirror of it."""
        return int(self.L.dawn_conv3x3_form(C.byref(d)))

    def ln_inline_ok(self, rows: int, N: int, C0: int, C1: int = 0) -> bool:
        """May the projection compute the LayerNorm of its input rows itself (conv_gemm(ln_eps=...): the row-stationary
        split GEMM holds whole rows in registers), so that no statistics pass reads them first?"""
        pol = self.conv_policy          # 0 = shipped policy; A/B policies without the split kernels (0x1000 clear) or with the
        if pol and (not (pol & 0x1000) or (pol & 0x20000)):          # row-stationary ones disabled (0x20000) take the statistics pass
            return False
        return bool(self.L.dawn_gemm1x1_ln_inline_ok(rows, N, C0, C1))

    def split_gemm_ok(self, rows: int, N: int, C0: int, C1: int = 0) -> bool:
        """Will a 1x1 projection of this shape run on a split-operand GEMM (which can apply the LayerNorm row statistics in
        its loader)?  Otherwise the caller materialises the normalised rows for the direct-to-LDS fp32 GEMM."""
        return bool(self.L.dawn_gemm1x1_split_ok(rows, N, C0, C1))

    # ------------------------------------------------------------------ GroupNorm / LayerNorm
    def conv_gn_part(self, rows_out: int, N: int, like: Tensor) -> Tensor:
        """Buffer for the GroupNorm partial sums a conv_gemm launch of this output shape emits (gn_part=...)."""
        return torch.empty(self.L.dawn_conv_gemm_nblocks(rows_out, N), 16, device=like.device, dtype=torch.float64)

    def gn_coeffs(self, x: Tensor, gamma: Tensor, beta: Tensor, film: Optional[Tuple[Tensor, Tensor]],
                  total_rows: int, eps: float = 1e-5, part: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        """Per-channel (a, b) with silu(x*a+b) == SiLU(FiLM(GroupNorm8(x))).  Statistics over all rows of the
        WHOLE clip: with a T-shard communicator the fp64 partial sums are all-reduced.  `part` = partial sums
        already produced by the conv epilogue (conv_gemm(gn_part=...)); otherwise a statistics pass runs."""
        rows, Cc = x.shape
        if part is not None and getattr(part, "dawn_ab", None) is not None:
            return part.dawn_ab                  # the conv launch that produced `part` finalised the coefficients itself (gn_fin)
        self._require(x, gamma, beta)
        sums = self.empty(16, like=x, dtype=torch.float64) if self.comm is not None else None
        s = self._stream()
        if part is None:
            nblk = max(1, min(1024, (rows * (Cc // 4) + 255) // 256 // 8))
            part = self.empty(nblk, 16, like=x, dtype=torch.float64)
            check(self.L.dawn_gn_partial(_p(x), rows, Cc, _ld(x), _p(part), nblk, s), "dawn_gn_partial")
        nblk = getattr(part, "dawn_rows", part.shape[0])    # rows written by the conv epilogue (or all of them)
        a = self.empty(Cc, like=x)
        b = self.empty(Cc, like=x)
        fs, fsh = (film if film is not None else (None, None))
        if self.comm is None:
            check(self.L.dawn_gn_reduce_finalize(_p(part), nblk, float(total_rows) * (Cc // 8), _p(gamma), _p(beta),
                                                 _p(fs), _p(fsh), Cc, eps, _p(a), _p(b), s), "dawn_gn_reduce_finalize")
            return a, b
        check(self.L.dawn_gn_reduce(_p(part), nblk, _p(sums), s), "dawn_gn_reduce")
        self.comm.all_reduce_sum(sums)
        check(self.L.dawn_gn_finalize(_p(sums), float(total_rows) * (Cc // 8), _p(gamma), _p(beta), _p(fs), _p(fsh),
                                      Cc, eps, _p(a), _p(b), s), "dawn_gn_finalize")
        return a, b

    def gn_apply_res(self, x: Tensor, a: Tensor, b: Tensor, res: Optional[Tensor], inplace: bool = False) -> Tensor:
        """out = SiLU(x * a + b) (+ res); inplace: written over x (the caller has no further use for the GroupNorm input --
        one tensor less at the allocator peak of long clips)."""
        rows, Cc = x.shape
        _need(x.is_contiguous() and (res is None or res.is_contiguous()), "gn_apply_res: x.is_contiguous() and (res is None or res.is_contiguous())")
        out = x if inplace else self.empty(rows, Cc, like=x)
        check(self.L.dawn_gn_apply_res(_p(x), _p(a), _p(b), _p(res), _p(out), rows, Cc, self._stream()),
              "dawn_gn_apply_res")
        return out

    def ln_rowstats(self, in0: Tensor, in1: Optional[Tensor] = None, eps: float = 1e-5) -> Tuple[Tensor, Tensor]:
        rows = in0.shape[0]
        self._require(in0, in1)
        mean = self.empty(rows, like=in0)
        rstd = self.empty(rows, like=in0)
        check(self.L.dawn_ln_rowstats(_p(in0), in0.shape[1], _ld(in0), _p(in1), 0 if in1 is None else in1.shape[1],
                                      _ld(in1), rows, eps, _p(mean), _p(rstd), self._stream()), "dawn_ln_rowstats")
        return mean, rstd

    def ln_rows(self, in0: Tensor, in1: Optional[Tensor] = None, eps: float = 1e-5) -> Tensor:
        """LayerNorm without gain of [in0|in1] rows, materialised (the gain lives in the next projection)."""
        rows = in0.shape[0]
        self._require(in0, in1)
        C = in0.shape[1] + (0 if in1 is None else in1.shape[1])
        xn = self.empty(rows, C, like=in0)
        check(self.L.dawn_ln_rows(_p(in0), in0.shape[1], _ld(in0), _p(in1), 0 if in1 is None else in1.shape[1],
                                  _ld(in1), rows, eps, _p(xn), self._stream()), "dawn_ln_rows")
        return xn

    # ------------------------------------------------------------------ cross attention
    def xattn_prep(self, kv: Tensor, k_scale: Tensor, null_kv: Tensor, kvtab: Tensor, branch: int,
                   nulltab: Tensor) -> None:
        self._require(kv, k_scale, null_kv, kvtab, nulltab)
        check(self.L.dawn_xattn_prep(_p(kv), kv.shape[0], _p(k_scale), _p(null_kv), _p(kvtab), branch, _p(nulltab),
                                     self._stream()), "dawn_xattn_prep")

    def xattn_core(self, q: Tensor, HW: int, kvtab: Tensor, nulltab: Tensor, q_scale: Tensor) -> Tensor:
        """In place on q (rows,192)."""
        _need(q.is_contiguous() and q.shape[1] == 192, "xattn_core: q.is_contiguous() and q.shape[1] == 192")
        self._require(q, kvtab, nulltab, q_scale)
        check(self.L.dawn_xattn_core(_p(q), _p(q), q.shape[0], HW, _p(kvtab), _p(nulltab), _p(q_scale),
                                     self._stream()), "dawn_xattn_core")
        return q

    def xattn_ln_sum(self, y3: Tensor, g3: Tensor, Co: int, eps: float = 1e-5) -> Tensor:
        rows = y3.shape[0]
        _need(y3.is_contiguous() and y3.shape[1] == 3 * Co, "xattn_ln_sum: y3.is_contiguous() and y3.shape[1] == 3 * Co")
        out = self.empty(rows, Co, like=y3)
        check(self.L.dawn_xattn_ln_sum(_p(y3), _p(g3), _p(out), rows, Co, eps, self._stream()), "dawn_xattn_ln_sum")
        return out

    @staticmethod
    def can_fuse_xattn(Cin: int, Co: int, C0: int, HW: int = 32) -> bool:
        return Co == 64 and Cin in (64, 128) and C0 % 8 == 0 and HW % 32 == 0

    def xattn_tables(self, kvtab: Tensor, nulltab: Tensor, q_scale: Tensor, wo, Co: int) -> Tensor:
        """Once per clip and conditioned block: (F,3,64+9*Co) = per (frame, branch) [D | u_0..u_7 | y0] such that
        softmax over [null, ctx] == sigmoid(-q.D/|q|) and to_out(o) == y0 + sum_h sigma_h u_h (include/dawn_hip.h)."""
        F = kvtab.shape[0]
        self._require(kvtab, nulltab, q_scale, *wo)
        xtab = torch.empty(F, 3, 64 + 9 * Co, device=kvtab.device, dtype=torch.float32)
        check(self.L.dawn_xattn_tables(_p(kvtab), _p(nulltab), _p(q_scale), _p(wo[0]), _p(wo[1]), _p(wo[2]), F, Co, _p(xtab),
                                       self._stream()), "dawn_xattn_tables")
        return xtab

    @staticmethod
    def can_fuse_xattn_out(Co: int, HW: int) -> bool:
        return Co % 32 == 0 and 32 <= Co <= 512 and HW % 4 == 0

    def xattn_sigma_out(self, q: Tensor, HW: int, xtab: Tensor, g3: Tensor, Co: int, eps: float = 1e-5,
                        gn: Optional[Tuple[Tensor, Tensor, Tensor]] = None, h1_over_c1: bool = False) -> Tensor:
        """q (rows,192) = raw to_q output -> h_cond (rows,Co): the 2-key attention, the three to_out projections, their
        LayerNorms and the branch sum in one pass (per-clip tables `xtab` from xattn_tables).  gn = (c1, a, b): returns the block's
        h1 = SiLU(c1*a + b) + h_cond instead (MT:473-476: no h_cond tensor, no GroupNorm-apply pass); h1_over_c1: written over c1
        (the epilogue reads an element of c1 and writes the same element of h1: include/dawn_hip.h)."""
        rows = q.shape[0]
        _need(q.is_contiguous() and q.shape[1] == 192 and xtab.is_contiguous() and xtab.shape[1:] == (3, 64 + 9 * Co), "xattn_sigma_out: q.is_contiguous() and q.shape[1] == 192 and xtab.is_contiguous() and xtab.shape[1:] == (3, 64 + 9 * Co)")
        _need(rows == xtab.shape[0] * HW, "xattn_sigma_out: rows == xtab.shape[0] * HW")
        self._require(q, xtab, g3)
        gx, ga, gb = gn if gn is not None else (None, None, None)
        _need(gx is None or (gx.is_contiguous() and tuple(gx.shape) == (rows, Co)), "xattn_sigma_out: gx is None or (gx.is_contiguous() and tuple(gx.shape) == (rows, Co))")
        out = gx if (h1_over_c1 and gx is not None) else self.empty(rows, Co, like=q)
        self._require(gx, ga, gb)
        check(self.L.dawn_xattn_sigma_out_h1(_p(q), rows, HW, _p(xtab), _p(g3), Co, eps, _p(gx), _p(ga), _p(gb), _p(out),
                                             self._stream()), "dawn_xattn_sigma_out")
        return out

    def xattn_layer_c64(self, x: Tensor, x2: Optional[Tensor], HW: int, wq: Tensor, wo, g3: Tensor, q_scale: Tensor,
                        kvtab: Tensor, nulltab: Tensor, eps: float = 1e-5, xtab: Optional[Tensor] = None,
                        wq_bf3: Optional[Tensor] = None, gn: Optional[Tuple[Tensor, Tensor, Tensor]] = None,
                        h1_over_c1: bool = False) -> Tensor:
        """h_cond (rows,64) = sum over the three branches of LN(to_out(attn(LN(x)))) in one launch.  The kernel reads
        the per-clip tables `xtab` (xattn_tables of kvtab / nulltab / q_scale / wo); built here if not supplied.
        `wq_bf3` (pack_bf3 of to_q) puts the Q projection on the bf16 matrix pipe (exact operand split).  h1_over_c1 (with gn): the
        block's h1 is written over c1 (see xattn_sigma_out)."""
        rows = x.shape[0]
        if xtab is None:
            xtab = self.xattn_tables(kvtab, nulltab, q_scale, wo, 64)
        self._require(x, x2, wq, g3, xtab)
        _need(xtab.is_contiguous() and xtab.shape[1:] == (3, 640) and rows == xtab.shape[0] * HW, "xattn_layer_c64: xtab.is_contiguous() and xtab.shape[1:] == (3, 640) and rows == xtab.shape[0] * HW")
        gx, ga, gb = gn if gn is not None else (None, None, None)       # (c1, a, b): write h1 = SiLU(c1*a + b) + h_cond instead of h_cond
        _need(gx is None or (gx.is_contiguous() and tuple(gx.shape) == (rows, 64)), "xattn_layer_c64: gx is None or (gx.is_contiguous() and tuple(gx.shape) == (rows, 64))")
        out = gx if (h1_over_c1 and gx is not None) else self.empty(rows, 64, like=x)
        self._require(gx, ga, gb)
        check(self.L.dawn_xattn_layer_c64_h1(_p(x), x.shape[1], _ld(x), _p(x2), 0 if x2 is None else x2.shape[1], _ld(x2),
                                             rows, HW, _p(wq), _p(wq_bf3), _p(g3), _p(xtab), eps, _p(gx), _p(ga), _p(gb), _p(out),
                                             self._stream()), "dawn_xattn_layer_c64")
        return out

    # ------------------------------------------------------------------ attention cores
    def temporal_attn(self, qkv: Tensor, Fext: int, HW: int, q0: int, Fq: int, win: int, rcos: Tensor, rsin: Tensor,
                      band: Tensor) -> Tensor:
        _need(qkv.is_contiguous() and qkv.shape == (Fext * HW, 768), "temporal_attn: qkv.is_contiguous() and qkv.shape == (Fext * HW, 768)")
        self._require(qkv, rcos, rsin, band)
        out = self.empty(Fq * HW, 256, like=qkv)
        check(self.L.dawn_temporal_attn_ex(_p(qkv), Fext, HW, q0, Fq, win, _p(rcos), _p(rsin), _p(band), _p(out),
                                           self.temporal_attn_flags, self._stream()), "dawn_temporal_attn")
        return out

    @staticmethod
    def can_fuse_temporal(C: int, Fext: int, Fq: int, win: int) -> bool:
        return C == 64 and Fext <= 288 and Fq <= 256 and win <= 48

    SEG_QUERIES = 120     # queries per launch of the segmented form: 120 + 2 x 40 halo rows = the 200-row LDS budget of WMODE 3

    @staticmethod
    def can_fuse_temporal_segmented(C: int, win: int) -> bool:
        return C == 64 and win <= 40

    def temporal_layer_c64_segmented(self, x: Tensor, Fext: int, HW: int, q0: int, Fq: int, win: int, wqkv: Tensor,
                                     wout: Tensor, rcos: Tensor, rsin: Tensor, band: Tensor, eps: float = 1e-5,
                                     wqkv_bf3: Optional[Tensor] = None, wout_bf3p: Optional[Tensor] = None,
                                     out: Optional[Tensor] = None, segments=None) -> Tensor:
        """The fused 64-channel layer for frame buffers longer than one launch holds in LDS (long clips, T-shard shards with
        halos): the query range is cut into segments of <= 120 frames and every segment is ONE launch of the fused kernel
        on the row window [a - win, b + win) of the same buffer (rows are frame-major, so a window is a contiguous view).
        Attention is window-local and rotary positions only matter relatively, so the segments are independent; the price
        is re-projecting K / V of the 2*win overlap rows per segment -- instead of writing and re-reading a (rows, 768)
        qkv tensor (21 -> ~10 MB per frame of peak memory on long clips).  `segments` = explicit [(a, b)] query ranges
        (buffer frame indices; the T-shard path runs the halo-free interior segments first)."""
        _need(x.is_contiguous() and x.shape == (Fext * HW, 64), "temporal_layer_c64_segmented: x.is_contiguous() and x.shape == (Fext * HW, 64)")
        if out is None:
            out = self.empty(Fq * HW, 64, like=x)
        if segments is None:
            segments = [(a, min(a + self.SEG_QUERIES, q0 + Fq)) for a in range(q0, q0 + Fq, self.SEG_QUERIES)]
        for a, b in segments:
            r0, r1 = max(0, a - win), min(Fext, b + win)
            self.temporal_layer_c64(x[r0 * HW:r1 * HW], r1 - r0, HW, a - r0, b - a, win, wqkv, wout, rcos, rsin, band, eps,
                                    wqkv_bf3=wqkv_bf3, wout_bf3p=wout_bf3p, out=out[(a - q0) * HW:(b - q0) * HW])
        return out

    def temporal_layer_c64(self, x: Tensor, Fext: int, HW: int, q0: int, Fq: int, win: int, wqkv: Tensor,
                           wout: Tensor, rcos: Tensor, rsin: Tensor, band: Tensor, eps: float = 1e-5,
                           wqkv_bf3: Optional[Tensor] = None, wout_bf3p: Optional[Tensor] = None,
                           out: Optional[Tensor] = None) -> Tensor:
        """out = x[q0:q0+Fq] + to_out(attn(LayerNorm(x))) for 64-channel levels, one kernel."""
        _need(x.is_contiguous() and x.shape == (Fext * HW, 64), "temporal_layer_c64: x.is_contiguous() and x.shape == (Fext * HW, 64)")
        self._require(x, wqkv, wout, rcos, rsin, band)
        if out is None:
            out = self.empty(Fq * HW, 64, like=x)
        _need(out.is_contiguous() and out.shape == (Fq * HW, 64), "temporal_layer_c64: out.is_contiguous() and out.shape == (Fq * HW, 64)")
        ev = None
        if getattr(self, "prof_layers", None) is not None and self.prof_on:      # bench.py: HIP events around the launch (sampled steps only)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        check(self.L.dawn_temporal_layer_c64_ex(_p(x), Fext, HW, q0, Fq, win, _p(wqkv), _p(wqkv_bf3), _p(wout), _p(wout_bf3p),
                                                _p(rcos), _p(rsin), _p(band), eps, _p(out), self.temporal_flags,
                                                self._stream()),
              "dawn_temporal_layer_c64")
        if ev is not None:
            ev[1].record()
            self.prof_layers.append(("temporal_layer_c64", (Fext, HW, q0, Fq, win, wqkv_bf3 is not None and wout_bf3p is not None), ev[0], ev[1]))
        return out

    def sla(self, qkv: Tensor, F: int, HW: int) -> Tensor:
        _need(qkv.is_contiguous() and qkv.shape == (F * HW, 768), "sla: qkv.is_contiguous() and qkv.shape == (F * HW, 768)")
        ctx = self.empty(F, 8, 32, 32, like=qkv)
        out = self.empty(F * HW, 256, like=qkv)
        s = self._stream()
        check(self.L.dawn_sla_context(_p(qkv), F, HW, _p(ctx), s), "dawn_sla_context")
        check(self.L.dawn_sla_apply(_p(qkv), _p(ctx), F, HW, _p(out), s), "dawn_sla_apply")
        return out

    def sla_layer_c64(self, x: Tensor, F: int, HW: int, wqkv: Tensor, wout: Tensor, bias: Tensor,
                      eps: float = 1e-5, wqkv_bf3: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
        """out = x + to_out(linear_attention(LayerNorm(x))) for 64-channel levels (two kernels, no qkv tensor)."""
        _need(x.is_contiguous() and x.shape == (F * HW, 64), "sla_layer_c64: x.is_contiguous() and x.shape == (F * HW, 64)")
        self._require(x, wqkv, wout, bias)
        ws = self.empty(self.L.dawn_sla_ws_floats(F, HW, int(wqkv_bf3 is not None)), like=x)
        if out is None:
            out = self.empty(F * HW, 64, like=x)
        _need(out.is_contiguous() and out.shape == (F * HW, 64), "sla_layer_c64: out.is_contiguous() and out.shape == (F * HW, 64)")
        check(self.L.dawn_sla_layer_c64(_p(x), F, HW, _p(wqkv), _p(wqkv_bf3), _p(wout), _p(bias), eps, _p(ws), _p(out),
                                        self._stream()), "dawn_sla_layer_c64")
        return out

    def frame_attn(self, qkv: Tensor, F: int, N: int) -> Tensor:
        _need(qkv.is_contiguous() and qkv.shape == (F * N, 768), "frame_attn: qkv.is_contiguous() and qkv.shape == (F * N, 768)")
        out = self.empty(F * N, 256, like=qkv)
        check(self.L.dawn_frame_attn(_p(qkv), F, N, _p(out), self._stream()), "dawn_frame_attn")
        return out

    # ------------------------------------------------------------------ boundary ops
    def init_conv_x(self, x: Tensor, w3: Tensor, fea_pre: Tensor, F: int, h: int, w: int, Co: int,
                    frames: Optional[Tuple[int, int]] = None, out: Optional[Tensor] = None) -> Tensor:
        """frames = (fa, fb): only that frame range of the (3, F, h, w) latent -> ((fb - fa)*h*w, Co) rows (T-shard: edge frames first)."""
        _need(x.is_contiguous() and x.shape == (3, F, h, w) and x.dtype == torch.float32, "init_conv_x: x.is_contiguous() and x.shape == (3, F, h, w) and x.dtype == torch.float32")     # (the frame offset below is in fp32 elements)
        self._require(x, w3, fea_pre, out)
        fa, fb = frames if frames is not None else (0, F)
        if out is None:
            out = self.empty((fb - fa) * h * w, Co, like=x)
        _need(out.is_contiguous() and out.shape == ((fb - fa) * h * w, Co), "init_conv_x: out.is_contiguous() and out.shape == ((fb - fa) * h * w, Co)")
        check(self.L.dawn_init_conv_x_ex(x.data_ptr() + fa * h * w * 4, F * h * w, _p(w3), _p(fea_pre), fb - fa, h, w, Co, _p(out),
                                         self._stream()), "dawn_init_conv_x")
        return out

    def head_out(self, hg: Optional[Tensor], ho: Optional[Tensor], wg: Tensor, bg: Tensor, wo: Tensor, bo: Tensor,
                 out: Optional[Tensor] = None) -> Tensor:
        """eps (3, rows): rows 0-1 from hg, row 2 from ho.  One of them may be None: only the other head's rows of `out` are written
        (long clips run the heads one after the other)."""
        ref = hg if hg is not None else ho
        rows, Co = ref.shape
        _need((hg is None or hg.is_contiguous()) and (ho is None or ho.is_contiguous()) and (out is not None or (hg is not None and ho is not None)), "head_out: (hg is None or hg.is_contiguous()) and (ho is None or ho.is_contiguous()) and (out is not None or (hg is not None and ho is not None))")
        if out is None:
            out = self.empty(3, rows, like=ref)
        check(self.L.dawn_head_out(_p(hg), _p(ho), _p(wg), _p(bg), _p(wo), _p(bo), rows, Co, _p(out),
                                   self._stream()), "dawn_head_out")
        return out

    def linear(self, x: Tensor, W: Tensor, bias: Optional[Tensor], act_in: int = 0,
               out: Optional[Tensor] = None) -> Tensor:
        M, K = x.shape
        N = W.shape[0]
        _need(W.is_contiguous() and W.shape[1] == K, "linear: W.is_contiguous() and W.shape[1] == K")
        self._require(x, W, bias, out)
        if out is None:
            out = self.empty(M, N, like=x)
        check(self.L.dawn_linear(_p(x), M, K, _ld(x), _p(W), _p(bias), N, act_in, _p(out), _ld(out), self._stream()),
              "dawn_linear")
        return out

    def sinusoidal(self, t: float, freqs: Tensor) -> Tensor:
        dim = 2 * freqs.numel()
        out = self.empty(1, dim, like=freqs)
        check(self.L.dawn_sinusoidal(float(t), dim, _p(freqs), _p(out), self._stream()), "dawn_sinusoidal")
        return out

    # ------------------------------------------------------------------ sampler
    def ddim_x0(self, x: Tensor, eps: Tensor, recip: float, recipm1: float) -> Tuple[Tensor, Tensor]:
        n = x.numel()
        _need(x.is_contiguous() and eps.is_contiguous(), "ddim_x0: x.is_contiguous() and eps.is_contiguous()")
        x0 = torch.empty_like(x)
        # selection scratch [hist1 2048 | hist2 1024 | hist3 1024 | state 4 | hmin 4]: ONE buffer per (device, stream), reset by two
        # stream-ordered fills per step (no allocation, no torch fill kernels inside the DDIM loop)
        key = (x.device.index, self._stream())
        ws = self._sel_ws.get(key)
        if ws is None:
            ws = self._sel_ws[key] = torch.empty(2048 + 1024 + 1024 + 8, device=x.device, dtype=torch.int32)
        check(self.L.dawn_select_ws_reset(_p(ws), self._stream()), "dawn_select_ws_reset")
        hist = ws[:2048]
        check(self.L.dawn_ddim_x0(_p(x), _p(eps), recip, recipm1, n, _p(x0), _p(hist), self._stream()), "dawn_ddim_x0")
        return x0, hist

    @staticmethod
    def quantile_rank(n_total: int, q: float) -> Tuple[int, float]:
        """(lower order statistic, interpolation weight) of the q-quantile of n_total values (MT:1186-1190).
        Up to 2^24 elements this is `torch.quantile`'s own arithmetic: the rank q*(n-1) is formed in the input dtype,
        fp32.  Above 2^24 torch refuses (no reference behaviour exists: the reference cannot sample such clips) and fp32
        cannot even represent n-1, so the rank is exact: floor / fraction of q*(n-1) in fp64 (the exact
        linearly interpolated order statistic)."""
        import numpy as np
        if n_total <= (1 << 24):
            pos = np.float32(q) * np.float32(n_total - 1)
            lo = int(np.floor(pos))
            return lo, float(np.float32(pos) - np.float32(lo))
        pos = np.float64(q) * np.float64(n_total - 1)
        lo = int(np.floor(pos))
        return lo, float(pos - np.float64(lo))

    def quantile_threshold(self, x0: Tensor, hist1: Tensor, n_total: int, q: float = 0.9) -> Tensor:
        """s = max(1, torch.quantile(|x0|, q)) over the WHOLE clip (histograms all-reduced when T-sharded).
        Returns a 2-float device tensor [s, raw quantile]."""
        lo, weight = self.quantile_rank(n_total, q)
        s = self._stream()
        n = x0.numel()
        ws = self._sel_ws.get((x0.device.index, s))
        if ws is None or hist1.data_ptr() != ws.data_ptr():          # histogram from elsewhere (tests): private scratch
            ws = torch.empty(2048 + 1024 + 1024 + 8, device=x0.device, dtype=torch.int32)
            check(self.L.dawn_select_ws_reset(_p(ws), s), "dawn_select_ws_reset")
        state = ws[4096:4100]
        if self.comm is not None:
            self.comm.all_reduce_sum(hist1)
        check(self.L.dawn_select_scan(_p(hist1), 2048, lo, _p(state), 1, s), "dawn_select_scan")
        for p in (2, 3):
            h = ws[2048 + (p - 2) * 1024:2048 + (p - 1) * 1024]
            check(self.L.dawn_select_hist(_p(x0), n, _p(state), p, _p(h), s), "dawn_select_hist")
            if self.comm is not None:
                self.comm.all_reduce_sum(h)
            check(self.L.dawn_select_scan(_p(h), 1024, 0, _p(state), p, s), "dawn_select_scan")
        hmin = ws[4100:4104]
        check(self.L.dawn_select_hist(_p(x0), n, _p(state), 4, _p(hmin), s), "dawn_select_hist")
        if self.comm is not None:
            self.comm.all_reduce_min(hmin)
        out = torch.empty(2, device=x0.device, dtype=torch.float32)
        check(self.L.dawn_select_finalize(_p(state), _p(hmin), weight, _p(out), s), "dawn_select_finalize")
        return out

    def ddim_update(self, x0: Tensor, eps: Tensor, s: Tensor, noise: Optional[Tensor], sqrt_alpha_next: float,
                    c: float, sigma: float) -> Tensor:
        x = torch.empty_like(x0)
        _need(noise is None or noise.is_contiguous(), "ddim_update: noise is None or noise.is_contiguous()")
        check(self.L.dawn_ddim_update(_p(x0), _p(eps), _p(s), _p(noise), sqrt_alpha_next, c, sigma, x0.numel(),
                                      _p(x), self._stream()), "dawn_ddim_update")
        return x

    def cfg_combine(self, e_null: Tensor, e_cond: Tensor, scale: float) -> Tensor:
        out = torch.empty_like(e_cond)
        check(self.L.dawn_cfg_combine(_p(e_null), _p(e_cond), float(scale), e_cond.numel(), _p(out), self._stream()),
              "dawn_cfg_combine")
        return out

    def philox_normal(self, Cc: int, F: int, f0: int, Ftotal: int, hw: int, seed: int, stream_id: int,
                      device) -> Tensor:
        out = torch.empty(Cc, F, hw, device=device, dtype=torch.float32)
        check(self.L.dawn_philox_normal(_p(out), Cc, F, f0, Ftotal, hw, seed, stream_id, self._stream()),
              "dawn_philox_normal")
        return out

    # ------------------------------------------------------------------ LFG flow decode
    def affine_act(self, x: Tensor, a: Tensor, b: Tensor, act: int = 1) -> Tensor:
        """out = act(x*a[c] + b[c]) on (rows, C); act 1 = ReLU (eval-mode BatchNorm + ReLU, UTIL:83-88)."""
        rows, Cc = x.shape
        self._require(x, a, b)
        out = self.empty(rows, Cc, like=x)
        check(self.L.dawn_affine_act(_p(x), _ld(x), _p(a), _p(b), act, _p(out), rows, Cc, self._stream()),
              "dawn_affine_act")
        return out

    def bn_relu_pool2(self, x: Tensor, a: Tensor, b: Tensor, F: int, H: int, W: int) -> Tensor:
        """(F*H*W, C) -> (F*H/2*W/2, C): AvgPool2x2(ReLU(x*a+b))  (DownBlock2d tail, UTIL:129-133)."""
        Cc = x.shape[1]
        _need(x.is_contiguous() and x.shape[0] == F * H * W, "bn_relu_pool2: x.is_contiguous() and x.shape[0] == F * H * W")
        self._require(x, a, b)
        out = self.empty(F * (H // 2) * (W // 2), Cc, like=x)
        check(self.L.dawn_bn_relu_pool2(_p(x), _p(a), _p(b), _p(out), F, H, W, Cc, self._stream()), "dawn_bn_relu_pool2")
        return out

    def warp_blend(self, skip: Tensor, Hs: int, Ws: int, grid: Tensor, conf: Tensor, prev: Optional[Tensor] = None,
                   prev_ab: Optional[Tuple[Tensor, Tensor]] = None, up2: bool = False) -> Tensor:
        """Generator.apply_optical (GEN:71-90) for T frames against the clip's single skip map (Hs*Ws, C).
        grid (2,T,h,w) view (planes may be strided: a frame range of a longer clip), conf (T,h,w) contiguous."""
        Cc = skip.shape[1]
        _, T, h, w = grid.shape
        _need(skip.is_contiguous() and skip.shape[0] == Hs * Ws and conf.is_contiguous() and conf.shape == (T, h, w), "warp_blend: skip.is_contiguous() and skip.shape[0] == Hs * Ws and conf.is_contiguous() and conf.shape == (T, h, w)")
        _need(grid.stride(3) == 1 and grid.stride(2) == w and grid.stride(1) == h * w, "warp_blend: grid.stride(3) == 1 and grid.stride(2) == w and grid.stride(1) == h * w")
        _need(prev is None or (prev.is_contiguous() and prev.shape == (T * Hs * Ws, Cc)), "warp_blend: prev is None or (prev.is_contiguous() and prev.shape == (T * Hs * Ws, Cc))")
        self._require(skip, grid, conf, prev)
        k = 2 if up2 else 1
        out = self.empty(T * Hs * k * Ws * k, Cc, like=skip)
        pa, pb = prev_ab if prev_ab is not None else (None, None)
        check(self.L.dawn_warp_blend(_p(skip), Hs, Ws, Cc, _p(grid), grid.stride(0), _p(conf), T, h, w, _p(prev), _p(pa),
                                     _p(pb), 1 if up2 else 0, _p(out), self._stream()), "dawn_warp_blend")
        return out

    def final_conv_blend(self, x: Tensor, H: int, W: int, w7: Tensor, bias3: Tensor, src: Tensor, grid: Tensor,
                         conf: Tensor, out_vid: Tensor, warped_vid: Tensor) -> None:
        """Generator.final + sigmoid + last apply_optical + `deformed` (GEN:152, 163-167).  x (T*H*W, C);
        src (3,H,W); out_vid / warped_vid: (3,T,H,W) views of the clip-sized outputs (frame range of (3,Ttot,H,W))."""
        _, T, h, w = grid.shape
        Cc = x.shape[1]
        _need(x.is_contiguous() and x.shape[0] == T * H * W and src.is_contiguous() and src.shape == (3, H, W), "final_conv_blend: x.is_contiguous() and x.shape[0] == T * H * W and src.is_contiguous() and src.shape == (3, H, W)")
        _need(grid.stride(3) == 1 and grid.stride(2) == w and grid.stride(1) == h * w and conf.is_contiguous(), "final_conv_blend: grid.stride(3) == 1 and grid.stride(2) == w and grid.stride(1) == h * w and conf.is_contiguous()")
        for o in (out_vid, warped_vid):
            _need(o.shape == (3, T, H, W) and o.stride(3) == 1 and o.stride(2) == W and o.stride(1) == H * W, "final_conv_blend: o.shape == (3, T, H, W) and o.stride(3) == 1 and o.stride(2) == W and o.stride(1) == H * W")
        _need(out_vid.stride(0) == warped_vid.stride(0), "final_conv_blend: out_vid.stride(0) == warped_vid.stride(0)")
        self._require(x, w7, bias3, src, grid, conf, out_vid, warped_vid)
        check(self.L.dawn_final_conv_blend(_p(x), T, H, W, Cc, _p(w7), _p(bias3), _p(src), _p(grid), grid.stride(0),
                                           _p(conf), h, w, _p(out_vid), _p(warped_vid), out_vid.stride(0),
                                           self._stream()), "dawn_final_conv_blend")

    # ------------------------------------------------------------------ frame egress
    def frames_to_u8(self, vid: Tensor, mean=(0.0, 0.0, 0.0), bgr: bool = False) -> Tensor:
        """(3,T,H,W) fp32 in [0,1] -> (T,H,W,3) uint8 with `_process_output_frame`'s arithmetic (UVG:533-548):
        trunc(clip(x + mean/255, 0, 1) * 255); `bgr=True` = the cv2 channel order."""
        _, T, H, W = vid.shape
        _need(vid.shape[0] == 3 and vid.stride(3) == 1 and vid.stride(2) == W and vid.stride(1) == H * W, "frames_to_u8: vid.shape[0] == 3 and vid.stride(3) == 1 and vid.stride(2) == W and vid.stride(1) == H * W")
        self._require(vid)
        out = torch.empty(T, H, W, 3, device=vid.device, dtype=torch.uint8)
        m = [float(x) / 255.0 for x in mean]
        check(self.L.dawn_frames_to_u8(_p(vid), vid.stride(0), T * H * W, m[0], m[1], m[2], 1 if bgr else 0, _p(out),
                                       self._stream()), "dawn_frames_to_u8")
        return out

    # ------------------------------------------------------------------ HuBERT audio features
    def wave_normalize(self, x: Tensor) -> Tensor:
        """Wav2Vec2FeatureExtractor(do_normalize): (x - mean) / sqrt(var + 1e-7) over the utterance."""
        _need(x.is_contiguous() and x.dim() == 1, "wave_normalize: x.is_contiguous() and x.dim() == 1")
        self._require(x)
        st = torch.empty(2, dtype=torch.float64, device=x.device)
        out = torch.empty_like(x)
        check(self.L.dawn_wave_normalize(_p(x), x.numel(), _p(st), _p(out), self._stream()), "dawn_wave_normalize")
        return out

    def hubert_conv0(self, x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int) -> Tensor:
        """Conv1d(1, C, k, stride) of the waveform -> (T0, C) rows; w (C, k)."""
        Cc, k = w.shape
        _need(x.is_contiguous() and w.is_contiguous(), "hubert_conv0: x.is_contiguous() and w.is_contiguous()")
        self._require(x, w, bias)
        T0 = (x.numel() - k) // stride + 1
        out = self.empty(T0, Cc, like=x)
        check(self.L.dawn_hubert_conv0(_p(x), x.numel(), _p(w), _p(bias), Cc, k, stride, _p(out), self._stream()),
              "dawn_hubert_conv0")
        return out

    def ln_affine_act(self, x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, act: int = 0) -> Tensor:
        """LayerNorm over the channels of every row, affine, act 0 none / 2 exact GELU."""
        _need(x.is_contiguous(), "ln_affine_act: x.is_contiguous()")
        self._require(x, gamma, beta)
        out = torch.empty_like(x)
        check(self.L.dawn_ln_affine_act(_p(x), x.shape[0], x.shape[1], _p(gamma), _p(beta), eps, act, _p(out), self._stream()),
              "dawn_ln_affine_act")
        return out

    def add_act(self, a: Optional[Tensor], b: Tensor, act: int = 0, out: Optional[Tensor] = None) -> Tensor:
        """out = a + act(b) (a may be None); act 2 = exact GELU."""
        _need(b.is_contiguous() and (a is None or a.is_contiguous()), "add_act: b.is_contiguous() and (a is None or a.is_contiguous())")
        self._require(a, b)
        out = torch.empty_like(b) if out is None else out
        check(self.L.dawn_add_act(_p(a), _p(b), act, b.numel(), _p(out), self._stream()), "dawn_add_act")
        return out

    def attn64(self, qkv: Tensor, heads: int) -> Tensor:
        """qkv (T, 3*heads*64) = [q | k | v] -> softmax(q k^T / 8) v per head, (T, heads*64)."""
        T = qkv.shape[0]
        _need(qkv.is_contiguous() and qkv.shape[1] == 3 * heads * 64, "attn64: qkv.is_contiguous() and qkv.shape[1] == 3 * heads * 64")
        self._require(qkv)
        out = self.empty(T, heads * 64, like=qkv)
        check(self.L.dawn_attn64(_p(qkv), T, heads, _p(out), self._stream()), "dawn_attn64")
        return out

    def attn_bias32(self, q: Tensor, k: Tensor, v: Tensor, heads: int, bias: Optional[Tensor], rcos: Optional[Tensor],
                    rsin: Optional[Tensor], scale: float) -> Tensor:
        """PBnet decoder attention: q (Tq, >= heads*32), k / v (Tk, ...) -- column slices allowed --, bias
        (heads, Tq, Tk) additive, rotary tables (>= max(Tq, Tk), nrot) for the first 2*nrot features of every head."""
        Tq, Tk = q.shape[0], k.shape[0]
        for t in (q, k, v):
            _need(t.stride(1) == 1 and t.shape[1] == heads * 32, "attn_bias32: t.stride(1) == 1 and t.shape[1] == heads * 32")
        self._require(q, k, v, bias, rcos, rsin)
        _need(bias is None or (bias.is_contiguous() and tuple(bias.shape) == (heads, Tq, Tk)), "attn_bias32: bias is None or (bias.is_contiguous() and tuple(bias.shape) == (heads, Tq, Tk))")
        nrot = 0 if rcos is None else rcos.shape[1]
        _need(rcos is None or (rcos.is_contiguous() and rsin.is_contiguous() and rcos.shape[0] >= max(Tq, Tk)), "attn_bias32: rcos is None or (rcos.is_contiguous() and rsin.is_contiguous() and rcos.shape[0] >= max(Tq, Tk))")
        out = self.empty(Tq, heads * 32, like=q)
        check(self.L.dawn_attn_bias32(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), Tq, Tk, heads, _p(bias), _p(rcos),
                                      _p(rsin), nrot, float(scale), _p(out), heads * 32, self._stream()), "dawn_attn_bias32")
        return out

    def interp_linear(self, y: Tensor, xi: Tensor) -> Tensor:
        """scipy interp1d(arange(n), y, kind='linear', axis=0)(xi) as float32; xi float64 positions on the device."""
        _need(y.is_contiguous() and xi.dtype == torch.float64 and xi.is_contiguous(), "interp_linear: y.is_contiguous() and xi.dtype == torch.float64 and xi.is_contiguous()")
        self._require(y, xi)
        out = self.empty(xi.numel(), y.shape[1], like=y)
        check(self.L.dawn_interp_linear(_p(y), y.shape[0], y.shape[1], _p(xi), xi.numel(), _p(out), self._stream()),
              "dawn_interp_linear")
        return out
