"""Op-level torch references with the SAME interface as dawn_pytorch_amd.ops.HipOps.

TEST INFRASTRUCTURE ONLY (see oracle/dawn_oracle.py header).  Two uses:
  * tests inject `RefOps` into the product's orchestration (unet_forward / sampler / tshard) to check the
    host logic + weight packing on CPU against the end-to-end oracle and the reference goldens;
  * `-m gpu` tests compare every HIP kernel against the method of the same name here.
Each method states the op's contract in the internal channels-last `(rows, C)` layout; the arithmetic
follows the reference lines cited in include/dawn_hip.h.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F_

Tensor = torch.Tensor


def _unpack(wp: Tensor) -> Tensor:
    K4, N, _ = wp.shape
    return wp.permute(0, 2, 1).reshape(K4 * 4, N)


class RefOps:
    name = "ref"

    def __init__(self, comm=None):
        self.comm = comm

    def with_comm(self, comm):
        return RefOps(comm)

    def fork_join(self, side, main):
        a = side()
        return a, main()

    def empty(self, *shape, like: Tensor, dtype=torch.float32) -> Tensor:
        return torch.zeros(*shape, device=like.device, dtype=dtype)

    # ------------------------------------------------------------------ conv / linear
    def conv_gemm(self, in0, w, N, *, F, Hi, Wi, Ho=None, Wo=None, KH=1, KW=1, stride=1, pad=0, mode=0, in1=None,
                  bias=None, row_stats=None, ch_ab=None, pro_act=0, pro_add=None, res=None, tr=None, out=None, w_bf3=None,
                  gn_part=None, ln_eps=0.0, w_wino=None, gn_fin=None, w_wino4=None):
        Ho = Hi if Ho is None else Ho
        Wo = Wi if Wo is None else Wo
        x = in0 if in1 is None else torch.cat((in0, in1), dim=1)
        Cin = x.shape[1]
        if ln_eps:                                   # LayerNorm (no gain) inside the projection
            mu = x.mean(dim=1, keepdim=True)
            x = (x - mu) * torch.rsqrt(((x - mu) ** 2).mean(dim=1, keepdim=True) + ln_eps)
        if row_stats is not None:
            x = (x - row_stats[0][:, None]) * row_stats[1][:, None]
        if ch_ab is not None:
            x = x * ch_ab[0][None, :] + ch_ab[1][None, :]
        if pro_act:
            x = F_.silu(x)
        if pro_add is not None:
            x = x + pro_add
        img = x.reshape(F, Hi, Wi, Cin).permute(0, 3, 1, 2)
        if mode == 0:
            wk = _unpack(w).reshape(KH, KW, Cin, N).permute(3, 2, 0, 1)
            y = F_.conv2d(img, wk, None, stride=stride, padding=pad)
            if y.shape[-2] >= Ho and y.shape[-1] >= Wo:
                y = y[:, :, :Ho, :Wo]            # the kernel computes the first (Ho, Wo) outputs only (HuBERT's SamePad drop)
        else:
            # 4 phase blocks (py,px) of 2x2 taps -> rebuild the ConvTranspose2d kernel (Cin, N, 4, 4)
            ksel = ((1, 3), (2, 0))
            wt = torch.zeros(Cin, N, 4, 4, device=x.device)
            for ph in range(4):
                py, px = ph >> 1, ph & 1
                blk = _unpack(w[ph]).reshape(2, 2, Cin, N)
                for ty in range(2):
                    for tx in range(2):
                        wt[:, :, ksel[py][ty], ksel[px][tx]] = blk[ty, tx]
            y = F_.conv_transpose2d(img, wt, None, stride=2, padding=1)
        assert y.shape[-2:] == (Ho, Wo), (y.shape, Ho, Wo)
        y = y.permute(0, 2, 3, 1).reshape(F * Ho * Wo, N)
        if bias is not None:
            y = y + bias
        if res is not None:
            y = y + res
        if tr is not None:
            y = y + F_.silu(tr[0] * tr[1][None, :] + tr[2][None, :])
        if gn_part is not None:
            yg = y.double().reshape(y.shape[0], 8, N // 8)
            gn_part.zero_()
            gn_part[0] = torch.stack((yg.sum(dim=(0, 2)), (yg * yg).sum(dim=(0, 2))), dim=1).reshape(16)
        if out is not None:
            out.copy_(y)
            return out
        return y.contiguous()

    def conv_gn_part(self, rows_out, N, like):
        return torch.zeros(3, 16, device=like.device, dtype=torch.float64)

    # ------------------------------------------------------------------ norms
    def gn_coeffs(self, x, gamma, beta, film, total_rows, eps=1e-5, part=None):
        rows, C = x.shape
        if part is not None:
            sums = part.sum(dim=0)
        else:
            xg = x.double().reshape(rows, 8, C // 8)
            sums = torch.stack((xg.sum(dim=(0, 2)), (xg * xg).sum(dim=(0, 2))), dim=1).reshape(16)
        if self.comm is not None:
            self.comm.all_reduce_sum(sums)
        sums = sums.reshape(8, 2)
        cnt = float(total_rows) * (C // 8)
        mean = sums[:, 0] / cnt
        var = (sums[:, 1] / cnt - mean * mean).clamp(min=0)
        rstd = (1.0 / torch.sqrt(var + eps)).float().repeat_interleave(C // 8)
        mu = mean.float().repeat_interleave(C // 8)
        a = rstd * gamma
        b = beta - mu * a
        if film is not None:
            sc = film[0] + 1.0
            a = a * sc
            b = b * sc + film[1]
        return a, b

    def gn_apply_res(self, x, a, b, res, inplace=False):
        y = F_.silu(x * a[None, :] + b[None, :])
        return y if res is None else y + res

    def ln_rowstats(self, in0, in1=None, eps=1e-5):
        x = in0 if in1 is None else torch.cat((in0, in1), dim=1)
        mean = x.mean(dim=1)
        var = x.var(dim=1, unbiased=False)
        return mean, 1.0 / torch.sqrt(var + eps)

    def ln_rows(self, in0, in1=None, eps=1e-5):
        x = in0 if in1 is None else torch.cat((in0, in1), dim=1)
        mean, rstd = self.ln_rowstats(in0, in1, eps)
        return (x - mean[:, None]) * rstd[:, None]

    # ------------------------------------------------------------------ cross attention
    def xattn_prep(self, kv, k_scale, null_kv, kvtab, branch, nulltab):
        Fn = kv.shape[0]
        k = kv[:, :64].reshape(Fn, 8, 8)
        k = F_.normalize(k, dim=-1) * k_scale
        kvtab[:, branch, :64] = k.reshape(Fn, 64)
        kvtab[:, branch, 64:] = kv[:, 64:]
        nulltab[branch, :8] = F_.normalize(null_kv[0], dim=-1) * k_scale
        nulltab[branch, 8:] = null_kv[1]

    def xattn_core(self, q, HW, kvtab, nulltab, q_scale):
        rows = q.shape[0]
        f = torch.arange(rows, device=q.device) // HW
        qq = q.reshape(rows, 3, 8, 8)
        qn = F_.normalize(qq, dim=-1) * q_scale[None, :, None, :]
        kc = kvtab[f][:, :, :64].reshape(rows, 3, 8, 8)
        vc = kvtab[f][:, :, 64:].reshape(rows, 3, 8, 8)
        kn = nulltab[:, :8][None, :, None, :]
        vn = nulltab[:, 8:][None, :, None, :]
        sn = (qn * kn).sum(-1) * 8.0
        sc = (qn * kc).sum(-1) * 8.0
        att = torch.stack((sn, sc), dim=-1).softmax(dim=-1)
        o = att[..., 0:1] * vn + att[..., 1:2] * vc
        q.copy_(o.reshape(rows, 192))
        return q

    def xattn_ln_sum(self, y3, g3, Co, eps=1e-5):
        rows = y3.shape[0]
        y = y3.reshape(rows, 3, Co)
        mean = y.mean(dim=-1, keepdim=True)
        var = y.var(dim=-1, unbiased=False, keepdim=True)
        return ((y - mean) * torch.rsqrt(var + eps) * g3[None]).sum(dim=1)

    @staticmethod
    def can_fuse_xattn(Cin, Co, C0, HW=32):
        return Co == 64 and Cin in (64, 128) and C0 % 8 == 0

    @staticmethod
    def ln_inline_ok(rows, N, C0, C1=0):
        return (C0 + C1) in (64, 128) or ((C0 + C1) % 128 == 0 and N <= 192)    # exercise both LayerNorm + projection forms on CPU

    @staticmethod
    def split_gemm_ok(rows, N, C0, C1=0):
        return True          # exercise the row-statistics form of the LayerNorm + projection on CPU

    @staticmethod
    def can_fuse_xattn_out(Co, HW):
        return Co % 32 == 0 and 32 <= Co <= 512 and HW % 4 == 0

    def xattn_sigma_out(self, q, HW, xtab, g3, Co, eps=1e-5, gn=None, h1_over_c1=False):
        """Reference of the one-pass kernel, evaluated from the per-clip tables (the tables themselves are checked
        against the definitions by test_xattn_tables, the whole chain against the original formulation by
        test_xattn_sigma_out_equals_unfused_chain)."""
        rows = q.shape[0]
        F = rows // HW
        qh = q.view(F, HW, 3, 8, 8)
        D = xtab[:, :, :64].view(F, 1, 3, 8, 8)
        n = qh.norm(dim=-1).clamp_min(1e-12)
        sig = 1.0 / (1.0 + torch.exp2((qh * D).sum(-1) / n))                                   # (F,HW,3,8)
        U = xtab[:, :, 64:64 + 8 * Co].view(F, 3, 8, Co)
        y0 = xtab[:, :, 64 + 8 * Co:]                                                          # (F,3,Co)
        y = y0[:, None] + torch.einsum("fpbh,fbhc->fpbc", sig, U)
        mean = y.mean(-1, keepdim=True)
        var = y.var(-1, unbiased=False, keepdim=True)
        out = ((y - mean) * torch.rsqrt(var + eps) * g3[None, None]).sum(2).reshape(rows, Co)
        return out if gn is None else F_.silu(gn[0] * gn[1] + gn[2]) + out        # h1 = SiLU(GN(c1)) + h_cond (MT:473-476)

    def xattn_tables(self, kvtab, nulltab, q_scale, wo, Co):
        """[D | u_0..u_7 | y0] per (frame, branch), written from the definitions (fp64)."""
        F = kvtab.shape[0]
        out = torch.zeros(F, 3, 64 + 9 * Co, dtype=torch.float64)
        for b in range(3):
            W = _unpack(wo[b]).double()                                   # (64, Co), k = 8 h + i
            kc, vc = kvtab[:, b, :64].double().view(F, 8, 8), kvtab[:, b, 64:].double().view(F, 8, 8)
            kn, vn = nulltab[b, :8].double(), nulltab[b, 8:].double()
            out[:, b, :64] = (q_scale[b].double()[None, None] * (kn[None, None] - kc) * (8.0 * math.log2(math.e))).reshape(F, 64)
            u = torch.einsum("fhi,hic->fhc", vc - vn[None, None], W.view(8, 8, Co))
            y0 = (vn.repeat(8)[:, None] * W).sum(0)
            out[:, b, 64:64 + 8 * Co] = u.reshape(F, 8 * Co)
            out[:, b, 64 + 8 * Co:] = y0[None]
        return out.float().to(kvtab.device)

    def xattn_layer_c64(self, x, x2, HW, wq, wo, g3, q_scale, kvtab, nulltab, eps=1e-5, xtab=None, wq_bf3=None, gn=None, h1_over_c1=False):
        """The ORIGINAL formulation (MT:516-559 op by op); `xtab` (the kernel's per-clip tables) is not used here, so the
        GPU test of the fused kernel also proves the table algebra."""
        rows = x.shape[0]
        stats = self.ln_rowstats(x, x2, eps)
        q = self.conv_gemm(x, wq, 192, in1=x2, row_stats=stats, F=rows, Hi=1, Wi=1)
        self.xattn_core(q, HW, kvtab, nulltab, q_scale)
        y3 = torch.zeros(rows, 192, device=x.device)
        for b in range(3):
            self.conv_gemm(q[:, 64 * b:64 * b + 64], wo[b], 64, F=rows, Hi=1, Wi=1, out=y3[:, 64 * b:64 * b + 64])
        out = self.xattn_ln_sum(y3, g3, 64, eps)
        return out if gn is None else F_.silu(gn[0] * gn[1] + gn[2]) + out

    # ------------------------------------------------------------------ attention cores
    def temporal_attn(self, qkv, Fext, HW, q0, Fq, win, rcos, rsin, band):
        x = qkv.reshape(Fext, HW, 3, 8, 32)
        q = x[:, :, 0].permute(1, 2, 0, 3) * 32 ** -0.5          # (HW, 8, Fext, 32)
        k = x[:, :, 1].permute(1, 2, 0, 3)
        v = x[:, :, 2].permute(1, 2, 0, 3)

        def rot(t):
            c, s = rcos[:Fext], rsin[:Fext]
            t1, t2 = t[..., 0::2], t[..., 1::2]
            return torch.stack((t1 * c - t2 * s, t2 * c + t1 * s), dim=-1).flatten(-2)

        q, k = rot(q)[:, :, q0:q0 + Fq], rot(k)
        i = torch.arange(q0, q0 + Fq, device=qkv.device)
        j = torch.arange(Fext, device=qkv.device)
        rel = j[None, :] - i[:, None]
        inside = rel.abs() <= win
        bias = band[(rel.clamp(-win, win) + win)].permute(2, 0, 1)           # (8, Fq, Fext)
        sim = torch.einsum("nhid,nhjd->nhij", q, k) + bias[None]
        sim = sim.masked_fill(~inside[None, None], float("-inf"))
        att = sim.softmax(dim=-1)
        o = torch.einsum("nhij,nhjd->nhid", att, v)                          # (HW, 8, Fq, 32)
        return o.permute(2, 0, 1, 3).reshape(Fq * HW, 256).contiguous()

    @staticmethod
    def can_fuse_temporal(C, Fext, Fq, win):
        return C == 64 and Fext <= 288 and Fq <= 256 and win <= 48

    SEG_QUERIES = 120

    @staticmethod
    def can_fuse_temporal_segmented(C, win):
        return C == 64 and win <= 40

    def temporal_layer_c64_segmented(self, x, Fext, HW, q0, Fq, win, wqkv, wout, rcos, rsin, band, eps=1e-5, wqkv_bf3=None,
                                     wout_bf3p=None, out=None, segments=None):
        """Same segmentation as HipOps (the CPU orchestration tests then also cover the row-window arithmetic)."""
        if out is None:
            out = torch.empty(Fq * HW, 64, dtype=x.dtype, device=x.device)
        if segments is None:
            segments = [(a, min(a + 120, q0 + Fq)) for a in range(q0, q0 + Fq, 120)]
        for a, b in segments:
            r0, r1 = max(0, a - win), min(Fext, b + win)
            out[(a - q0) * HW:(b - q0) * HW] = self.temporal_layer_c64(x[r0 * HW:r1 * HW], r1 - r0, HW, a - r0, b - a, win, wqkv,
                                                                       wout, rcos, rsin, band, eps)
        return out

    def temporal_layer_c64(self, x, Fext, HW, q0, Fq, win, wqkv, wout, rcos, rsin, band, eps=1e-5, wqkv_bf3=None, wout_bf3p=None,
                           out=None):
        """Composition of the unfused reference ops (what the fused kernel must equal)."""
        stats = self.ln_rowstats(x, None, eps)
        qkv = self.conv_gemm(x, wqkv, 768, row_stats=stats, F=Fext, Hi=1, Wi=HW)
        o = self.temporal_attn(qkv, Fext, HW, q0, Fq, win, rcos, rsin, band)
        return self.conv_gemm(o, wout, 64, res=x[q0 * HW:(q0 + Fq) * HW], F=Fq, Hi=1, Wi=HW)

    def sla(self, qkv, F, HW):
        x = qkv.reshape(F, HW, 3, 8, 32)
        q = x[:, :, 0].permute(0, 2, 3, 1)                                   # (F, 8, 32, HW)
        k = x[:, :, 1].permute(0, 2, 3, 1)
        v = x[:, :, 2].permute(0, 2, 3, 1)
        q = q.softmax(dim=-2) * 32 ** -0.5
        k = k.softmax(dim=-1)
        ctx = torch.einsum("fhdn,fhen->fhde", k, v)
        o = torch.einsum("fhde,fhdn->fhen", ctx, q)                          # (F, 8, 32, HW)
        return o.permute(0, 3, 1, 2).reshape(F * HW, 256).contiguous()

    def sla_layer_c64(self, x, F, HW, wqkv, wout, bias, eps=1e-5, wqkv_bf3=None, out=None):
        stats = self.ln_rowstats(x, None, eps)
        qkv = self.conv_gemm(x, wqkv, 768, row_stats=stats, F=F, Hi=1, Wi=HW)
        o = self.sla(qkv, F, HW)
        return self.conv_gemm(o, wout, 64, bias=bias, res=x, F=F, Hi=1, Wi=HW, out=out)

    def frame_attn(self, qkv, F, N):
        x = qkv.reshape(F, N, 3, 8, 32)
        q = x[:, :, 0].permute(0, 2, 1, 3) * 32 ** -0.5
        k = x[:, :, 1].permute(0, 2, 1, 3)
        v = x[:, :, 2].permute(0, 2, 1, 3)
        att = torch.einsum("fhid,fhjd->fhij", q, k).softmax(dim=-1)
        o = torch.einsum("fhij,fhjd->fhid", att, v)
        return o.permute(0, 2, 1, 3).reshape(F * N, 256).contiguous()

    # ------------------------------------------------------------------ boundary ops
    def init_conv_x(self, x, w3, fea_pre, F, h, w, Co, frames=None, out=None):
        fa, fb = frames if frames is not None else (0, F)
        wk = w3.reshape(7, 7, 3, Co).permute(3, 2, 0, 1)
        y = F_.conv2d(x[:, fa:fb].permute(1, 0, 2, 3), wk, None, padding=3)  # (fb - fa, Co, h, w)
        y = y.permute(0, 2, 3, 1).reshape(fb - fa, h * w, Co) + fea_pre[None]
        y = y.reshape((fb - fa) * h * w, Co).contiguous()
        if out is not None:
            out.copy_(y)
            return out
        return y

    def head_out(self, hg, ho, wg, bg, wo, bo, out=None):
        if hg is not None and ho is not None and out is None:
            return torch.cat((hg @ wg.t() + bg, ho @ wo.t() + bo), dim=1).t().contiguous()
        if out is None:
            out = torch.empty(3, (hg if hg is not None else ho).shape[0])
        if hg is not None:
            out[:2] = (hg @ wg.t() + bg).t()
        if ho is not None:
            out[2:] = (ho @ wo.t() + bo).t()
        return out

    def linear(self, x, W, bias, act_in=0, out=None):
        if act_in == 1:
            x = F_.silu(x)
        elif act_in == 2:
            x = F_.gelu(x)
        y = F_.linear(x, W, bias)
        if out is not None:
            out.copy_(y)
            return out
        return y

    def sinusoidal(self, t, freqs):
        e = torch.tensor([float(t)], dtype=torch.float32, device=freqs.device)[:, None] * freqs[None, :]
        return torch.cat((e.sin(), e.cos()), dim=-1)

    # ------------------------------------------------------------------ sampler
    def ddim_x0(self, x, eps, recip, recipm1):
        x0 = recip * x - recipm1 * eps
        bits = x0.abs().contiguous().view(torch.int32)
        hist = torch.bincount((bits >> 20).flatten().long(), minlength=2048).to(torch.int32)
        return x0, hist

    def quantile_threshold(self, x0, hist1, n_total, q=0.9):
        """Exact linear-interpolated order statistic (== torch.quantile for n <= 2^24), written via a
        full sort; with a communicator the shards' |x0| are gathered first."""
        v = x0.abs().flatten()
        if self.comm is not None:
            v = self.comm.all_gather_cat(v)
        if n_total <= (1 << 24):                       # torch.quantile's own rank arithmetic (input dtype = fp32)
            pos = np.float32(q) * np.float32(n_total - 1)
            lo = int(np.floor(pos))
            wgt = float(np.float32(pos) - np.float32(lo))
        else:                                          # torch refuses above 2^24: exact rank in fp64
            pos = np.float64(q) * np.float64(n_total - 1)
            lo = int(pos // 1)
            wgt = float(pos - lo)
        sv = torch.sort(v).values
        a, b = sv[lo], sv[min(lo + 1, sv.numel() - 1)]
        qv = torch.lerp(a, b, torch.tensor(wgt, dtype=torch.float32, device=v.device))
        return torch.stack((qv.clamp(min=1.0), qv))

    def ddim_update(self, x0, eps, s, noise, sqrt_alpha_next, c, sigma):
        sv = s[0]
        x = torch.minimum(torch.maximum(x0, -sv), sv) / sv * sqrt_alpha_next + c * eps
        if noise is not None:
            x = x + sigma * noise
        return x

    def cfg_combine(self, e_null, e_cond, scale):
        return e_null + (e_cond - e_null) * scale

    def philox_normal(self, Cc, F, f0, Ftotal, hw, seed, stream_id, device):
        """Bit-level restatement of the kernel's Philox4x32-10 + Box-Muller (numpy uint64 arithmetic)."""
        qpf = hw // 4
        c = np.arange(Cc, dtype=np.uint64)[:, None, None]
        f = (np.arange(F, dtype=np.uint64) + np.uint64(f0))[None, :, None]
        qd = np.arange(qpf, dtype=np.uint64)[None, None, :]
        gq = (c * np.uint64(Ftotal) + f) * np.uint64(qpf) + qd
        M32 = np.uint64(0xFFFFFFFF)
        c0, c1 = gq & M32, gq >> np.uint64(32)
        c2 = np.full_like(gq, stream_id)
        c3 = np.full_like(gq, 0x44415757)
        k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
        for _ in range(10):
            p0 = np.uint64(0xD2511F53) * c0
            p1 = np.uint64(0xCD9E8D57) * c2
            n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
            n1 = p1 & M32
            n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
            n3 = p0 & M32
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = (k0 + np.uint64(0x9E3779B9)) & M32
            k1 = (k1 + np.uint64(0xBB67AE85)) & M32
        two32 = np.float32(2.3283064365386963e-10)
        u = [((ci.astype(np.float32) + np.float32(0.5)) * two32) for ci in (c0, c1, c2, c3)]
        r0 = np.sqrt(np.float32(-2.0) * np.log(np.clip(u[0], 1e-10, 1.0)))
        r1 = np.sqrt(np.float32(-2.0) * np.log(np.clip(u[2], 1e-10, 1.0)))
        a0 = np.float32(6.283185307179586) * u[1]
        a1 = np.float32(6.283185307179586) * u[3]
        out = np.stack((r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)), axis=-1)
        return torch.from_numpy(out.reshape(Cc, F, hw).astype(np.float32)).to(device)

    # ------------------------------------------------------------------ LFG flow decode
    def affine_act(self, x, a, b, act=1):
        y = x * a + b
        return F_.relu(y) if act == 1 else y

    def bn_relu_pool2(self, x, a, b, F, H, W):
        Cc = x.shape[1]
        y = F_.relu(x * a + b).view(F, H, W, Cc).permute(0, 3, 1, 2)
        return F_.avg_pool2d(y, 2).permute(0, 2, 3, 1).reshape(-1, Cc)

    @staticmethod
    def _motion(grid, conf, Hs, Ws):
        """flow (T,Hs,Ws,2) and occlusion (T,1,Hs,Ws) at the skip's resolution (GEN:63-68, 81-82)."""
        _, T, h, w = grid.shape
        flow = grid.permute(1, 0, 2, 3)                       # (T,2,h,w)
        occ = conf.view(T, 1, h, w)
        if (h, w) != (Hs, Ws):
            flow = F_.interpolate(flow, size=(Hs, Ws), mode="bilinear")
            occ = F_.interpolate(occ, size=(Hs, Ws), mode="bilinear")
        return flow.permute(0, 2, 3, 1), occ

    def warp_blend(self, skip, Hs, Ws, grid, conf, prev=None, prev_ab=None, up2=False):
        Cc = skip.shape[1]
        T = grid.shape[1]
        flow, occ = self._motion(grid, conf, Hs, Ws)
        src = skip.view(1, Hs, Ws, Cc).permute(0, 3, 1, 2).expand(T, -1, -1, -1)
        out = F_.grid_sample(src, flow, mode="bilinear", padding_mode="zeros", align_corners=False) * occ
        if prev is not None:
            p = prev if prev_ab is None else F_.relu(prev * prev_ab[0] + prev_ab[1])
            out = out + p.view(T, Hs, Ws, Cc).permute(0, 3, 1, 2) * (1 - occ)
        if up2:
            out = F_.interpolate(out, scale_factor=2)
        return out.permute(0, 2, 3, 1).reshape(-1, Cc)

    def final_conv_blend(self, x, H, W, w7, bias3, src, grid, conf, out_vid, warped_vid):
        T = grid.shape[1]
        Cc = x.shape[1]
        wt = w7.view(7, 7, Cc // 4, 3, 4).permute(3, 2, 4, 0, 1).reshape(3, Cc, 7, 7)   # [tap][C/4][3][4] -> (3,C,7,7)
        y = torch.sigmoid(F_.conv2d(x.view(T, H, W, Cc).permute(0, 3, 1, 2), wt, bias3, padding=3))
        flow, occ = self._motion(grid, conf, H, W)
        wsrc = F_.grid_sample(src.view(1, 3, H, W).expand(T, -1, -1, -1), flow, mode="bilinear", padding_mode="zeros",
                              align_corners=False)
        warped_vid.copy_(wsrc.permute(1, 0, 2, 3))
        out_vid.copy_((wsrc * occ + y * (1 - occ)).permute(1, 0, 2, 3))

    # ------------------------------------------------------------------ frame egress
    def frames_to_u8(self, vid, mean=(0.0, 0.0, 0.0), bgr=False):
        """numpy restatement of UVG:533-548 applied to every frame: float32 frame, in-place += float64 mean/255,
        clip, *255 (float32), astype(uint8); cv2.cvtColor(RGB2BGR) = channel reversal."""
        frame = vid.permute(1, 2, 3, 0).detach().cpu().numpy().astype(np.float32).copy()   # (T,H,W,3)
        frame += np.array(mean) / 255.0
        frame = np.clip(frame, 0, 1)
        frame = (frame * 255).astype(np.uint8)
        if bgr:
            frame = frame[..., ::-1].copy()
        return torch.from_numpy(frame)

    # ------------------------------------------------------------------ HuBERT audio features
    def wave_normalize(self, x):
        x = x.float()
        return (x - x.mean()) / torch.sqrt(x.var(unbiased=False) + 1e-7)

    def hubert_conv0(self, x, w, bias, stride):
        y = F_.conv1d(x[None, None], w[:, None, :], bias, stride=stride)[0]          # (C, T0)
        return y.t().contiguous()

    def ln_affine_act(self, x, gamma, beta, eps=1e-5, act=0):
        y = F_.layer_norm(x, (x.shape[1],), gamma, beta, eps)
        return F_.gelu(y) if act == 2 else y

    def add_act(self, a, b, act=0, out=None):
        y = F_.gelu(b) if act == 2 else b
        y = y if a is None else a + y
        if out is not None:
            out.copy_(y)
            return out
        return y

    def attn64(self, qkv, heads):
        T_ = qkv.shape[0]
        q, k, v = (t.reshape(T_, heads, 64).transpose(0, 1) for t in qkv.chunk(3, dim=1))
        att = torch.softmax((q * 0.125) @ k.transpose(1, 2), dim=-1)
        return (att @ v).transpose(0, 1).reshape(T_, heads * 64).contiguous()

    def attn_bias32(self, q, k, v, heads, bias, rcos, rsin, scale):
        def split(t):
            return t.reshape(t.shape[0], heads, 32).transpose(0, 1)

        def rot(t):                                              # interleaved pairs of the first 2*nrot features, position = row
            if rcos is None:
                return t
            n, nr = t.shape[1], rcos.shape[1]
            c, s = rcos[:n].repeat_interleave(2, dim=1), rsin[:n].repeat_interleave(2, dim=1)
            tr, tp = t[..., :2 * nr], t[..., 2 * nr:]
            x = tr.reshape(*tr.shape[:-1], nr, 2)
            half = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(tr.shape)
            return torch.cat((tr * c + half * s, tp), dim=-1)
        qh, kh, vh = rot(split(q) * scale), rot(split(k)), split(v)
        sim = qh @ kh.transpose(1, 2)
        if bias is not None:
            sim = sim + bias
        sim = sim - sim.amax(dim=-1, keepdim=True)
        return (sim.softmax(dim=-1) @ vh).transpose(0, 1).reshape(q.shape[0], heads * 32).contiguous()

    def interp_linear(self, y, xi):
        from scipy.interpolate import interp1d
        f = interp1d(np.arange(y.shape[0]), y.cpu().numpy(), kind="linear", axis=0)
        return torch.from_numpy(f(xi.cpu().numpy()).astype(np.float32))
